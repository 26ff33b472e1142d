// vmaf_kernels.hip - hand-written gfx950 (MI355X / CDNA4) kernels for the VMAF v0.6.1 feature path.
//
// Numerics policy: every multiply and add in the stencils rounds separately (build with
// -ffp-contract=off) and the filter taps are summed in one of the reference's own two orders, chosen
// like the reference chooses (compute_vmaf's disable_avx flag, libvmaf.cpp:56-61):
//   ORDER_AVX    (default) the order of the reference's AVX path, which is what it runs on x86:
//                interior samples use four interleaved partial sums, border rows/columns the scalar
//                order (feature/src/common/convolution_avx.c:787-837, SURVEY.md note N1)
//   ORDER_SCALAR taps accumulated left to right everywhere (vif_tools.c:292-328, convolution.c:26-76)
// ADM has a single order in the reference (adm_tools.c:308-389).  Per-pixel values are therefore
// bit-identical to the reference in the same mode; only the final plane sums differ (fp64 tree here,
// fp32 row-then-total there).
//   ORDER_FMA    opt-in fast mode, not a reference order: sequential taps with fused multiply-adds
//                (VIF, motion and ADM's db2 taps); within the parity tolerances, not bit-identical.
//
// Kernels (one launch covers a whole batch of frames: blockIdx.z = frame):
//   vif_stat_kernel<FW>      fused: tile+halo of ref/dis -> LDS, vertical pass of {r,d,r^2,d^2,rd},
//                            horizontal pass, per-pixel VIF statistic, block-reduced num/den; from the same tile
//                            the next scale's input (blur with the next filter at even samples only)
//   vif_stat_avx_kernel<FW>  the same for ORDER_AVX: border-frame tiles (per-sample order) and interior tiles
//                            (pure vector order) in one launch
//   adm_scale_kernel         fused: db2 DWT (ref+dis), decouple, CSF, 3x3 masking threshold,
//                            masked cube sums over the centre window, LL bands to HBM
//   motion_kernel            5-tap blur of frame f and f-1 (or carried blurred plane), SAD
//   reduce_kernel         deterministic fp64 reduction of the per-block partials
//   svm_rbf_kernel        batched nu-SVR RBF prediction
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "vmaf_kernels.h"

// typed buffer load (buffer_load_format_x): the LLVM intrinsic bound by name, see load_px below
__device__ float llvm_amdgcn_raw_ptr_buffer_load_format_f32(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.ptr.buffer.load.format.f32");

namespace {

// ---------------------------------------------------------------- tables
// feature/src/vif_tools.c:82-87
__device__ constexpr float kVif17[17] = { 0.00745626912f, 0.0142655009f, 0.0250313189f, 0.0402820669f,
    0.0594526194f, 0.0804751068f, 0.0999041125f, 0.113746084f, 0.118773937f, 0.113746084f, 0.0999041125f,
    0.0804751068f, 0.0594526194f, 0.0402820669f, 0.0250313189f, 0.0142655009f, 0.00745626912f };
__device__ constexpr float kVif9[9] = { 0.0189780835f, 0.0558981746f, 0.120920904f, 0.192116052f, 0.224173605f,
    0.192116052f, 0.120920904f, 0.0558981746f, 0.0189780835f };
__device__ constexpr float kVif5[5] = { 0.054488685f, 0.244201347f, 0.402619958f, 0.244201347f, 0.054488685f };
__device__ constexpr float kVif3[3] = { 0.166378498f, 0.667243004f, 0.166378498f };
// feature/src/motion_tools.h:24-29
__device__ constexpr float kMotion5[5] = { 0.054488685f, 0.244201342f, 0.402619947f, 0.244201342f, 0.054488685f };
// feature/src/adm_tools.c:48-49
__device__ constexpr float kDb2Lo[4] = { 0.482962913144690f, 0.836516303737469f, 0.224143868041857f, -0.129409522550921f };
__device__ constexpr float kDb2Hi[4] = { -0.129409522550921f, -0.224143868041857f, 0.836516303737469f, -0.482962913144690f };

template <int FW> __device__ __forceinline__ constexpr float vif_tap(int k)
{
    return FW == 17 ? kVif17[k] : FW == 9 ? kVif9[k] : FW == 5 ? kVif5[k] : kVif3[k];
}
// MOTION selects FILTER_5_s (motion_tools.h), which differs from VIF's 5-tap table in the 8th digit
template <int FW, bool MOTION> __device__ __forceinline__ constexpr float ftap(int k) { return MOTION ? kMotion5[k] : vif_tap<FW>(k); }

constexpr int ORDER_SCALAR = 0, ORDER_AVX = 1, ORDER_FMA = 2;

// One output sample from p[0], p[st], .., p[(FW-1)*st], taps accumulated left to right.  The reference starts
// from 0.0f; starting from the first product gives the same bits (0.0f + x == x) except for the sign of a
// zero sum, which nothing downstream can see (every consumer squares, compares or adds it).
template <int FW, bool MOTION> __device__ __forceinline__ float taps_seq_at(const float *p, int st)
{
    float acc = ftap<FW, MOTION>(0) * p[0];
#pragma unroll
    for (int k = 1; k < FW; ++k) acc = acc + ftap<FW, MOTION>(k) * p[k * st];
    return acc;
}

// ORDER_FMA, the opt-in fast mode: the sequential order with every multiply-add fused (one rounding per tap instead of
// two).  NOT one of the reference's two orders: samples differ from them in the last bits (features by ~1e-6, inside the
// 1e-4 parity tolerance but not bit-identical), in exchange for half the VALU instructions of a convolution.
template <int FW, bool MOTION> __device__ __forceinline__ float taps_fma_at(const float *p, int st)
{
    float acc = ftap<FW, MOTION>(0) * p[0];
#pragma unroll
    for (int k = 1; k < FW; ++k) acc = __builtin_fmaf(ftap<FW, MOTION>(k), p[k * st], acc);
    return acc;
}

// Same sample in the interior order of the reference's AVX kernels (convolution_avx.c:225-270, :347-392,
// :408-437, generic :116-205): passes of up to 9 taps; inside a pass tap k goes to partial sum k%4, the pass
// value is (s0+s2)+(s1+s3); pass values are added in order (the first one to +0: see above).
template <int FW, bool MOTION> __device__ __forceinline__ float taps_avx_at(const float *p, int st)
{
    float total = 0.0f;
#pragma unroll
    for (int x = 0; x < FW; x += 9) {
        const int n = FW - x < 9 ? FW - x : 9;
        float sm[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (k < n) {
                const float prod = ftap<FW, MOTION>(x + k) * p[(x + k) * st];
                sm[k & 3] = (k < 4) ? prod : sm[k & 3] + prod;
            }
        }
        const float pass = (sm[0] + sm[2]) + (sm[1] + sm[3]);
        total = (x == 0) ? pass : total + pass;
    }
    return total;
}

// A sample whose order is decided per lane (ORDER_AVX: vector order in the interior, scalar order where the
// reference falls back to its scalar loops).  The branches are wave-uniform: a wave evaluates an order only
// if one of its lanes needs it, so interior waves pay for one.
template <int FW, bool MOTION, int ORDER> __device__ __forceinline__ float taps_sel(const float *p, int st, bool avx)
{
    if (ORDER == ORDER_SCALAR) return taps_seq_at<FW, MOTION>(p, st);
    if (ORDER == ORDER_FMA) return taps_fma_at<FW, MOTION>(p, st);
    float ra = 0.0f, rs = 0.0f;
    if (__any(avx)) ra = taps_avx_at<FW, MOTION>(p, st);
    if (__any(!avx)) rs = taps_seq_at<FW, MOTION>(p, st);
    return avx ? ra : rs;
}

// Where the reference's AVX path uses its vector kernels (convolution_avx.c:799-836):
//   vertical pass:   rows r <= i < h - r, columns j < floor8(w)
//   horizontal pass: columns r <= j < r + floor8(w) - ceil8(r + 1)
__host__ __device__ __forceinline__ bool avx_row_ok(int gy, int h, int r) { return gy >= r && gy < h - r; }
__host__ __device__ __forceinline__ bool avx_vcol_ok(int gx, int w) { return gx < (w & ~7); }
__host__ __device__ __forceinline__ int avx_hcol_end(int w, int r) { return r + (w & ~7) - ((r + 1 + 7) & ~7); }

// Mirror rule of every stencil on the path (vif_tools.c:301, adm_tools.c:241-248); the final
// clamp only matters for tiles hanging over the image, whose results are masked out.
__device__ __forceinline__ int mirror_idx(int i, int n)
{
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - i - 1 : i;
    return min(max(i, 0), n - 1);
}

// ---------------------------------------------------------------- XCD-aware tile order
// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so blocks b and b+8 share an L2.
// Tiles are numbered row-major; this bijection gives the blocks of one XCD a contiguous run of tiles, so the
// halo rows/columns that neighbouring tiles have in common are fetched into that L2 once instead of once per
// XCD.  Speed only - any placement is correct.  (grid.x = number of tiles, grid.z = frame)
__device__ __forceinline__ int xcd_tile_index(int b, int nt)
{
    const int q = nt >> 3, r = nt & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}
__device__ __forceinline__ int xcd_tile_index() { return xcd_tile_index(blockIdx.x, gridDim.x); }

// ---------------------------------------------------------------- block reduction (deterministic)
template <int K, int NWAVES>
__device__ __forceinline__ void block_reduce_store(double (&v)[K], double *red /* NWAVES*K doubles of LDS */, double *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0) red[wave * K + k] = x;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double s = 0;
#pragma unroll
        for (int wv = 0; wv < NWAVES; ++wv) s += red[wv * K + threadIdx.x];
        out[threadIdx.x] = s;
    }
}

// Same result layout, but through LDS instead of 6 x 2 x K cross-lane shuffles per thread: every thread parks its K
// values, K x 32 threads add 8 of them each in a fixed order and finish with a 32-lane shuffle tree.  Deterministic.
// buf: K * NT doubles of LDS (may alias data that is dead after the leading barrier).
template <int K, int NT>
__device__ __forceinline__ void block_reduce_store_lds(double (&v)[K], double *buf, double *out)
{
    static_assert(NT == 256 && K * 32 <= NT, "256 threads, at most 8 values");
    const int tid = threadIdx.x;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) buf[k * NT + tid] = v[k];
    __syncthreads();
    if (tid < K * 32) {
        const int k = tid >> 5, j = tid & 31;
        double s = buf[k * NT + j];
#pragma unroll
        for (int i = 1; i < NT / 32; ++i) s += buf[k * NT + j + 32 * i];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) s += __shfl_down(s, off, 32);
        if (j == 0) out[k] = s;
    }
}

// Buffer-addressed loads (cdna_hip_programming.md T8/T20): a 128-bit descriptor of one frame in SGPRs, a wave-uniform
// byte offset in an SGPR and the per-lane byte offset in one VGPR - no 64-bit per-lane address arithmetic.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t frame_rsrc(const float *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, int voffset_bytes, int soffset_bytes)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voffset_bytes, soffset_bytes, 0));
}

// One luma sample of the caller's picture as the float the reference's readers deliver (file_io.c:112-116, 159-163):
//   float          the planes of the read_frame contract, as they are
//   unsigned char  (float)px                       read_image_b2s
//   unsigned short (float)px / 4.0 = px * 0.25f    read_image_w2s (little-endian words, no masking to 10 bits); exact
// The kernels that read the picture (VIF scale 0, ADM scale 0, motion) are templated on the sample type, so raw 8/16-bit
// luma is widened in the load and never exists as fp32 planes in HBM.  x = sample index in the row, row offset in bytes.
// Integer samples come through TYPED buffer loads: the frame's descriptor names their format (8 or 16 bit, USCALED =
// unsigned integer -> float, exact) and the load path itself delivers (float)px - no conversion instruction in the kernel
// (buffer_load_format_x issues at the rate of buffer_load_ubyte / _ushort / _dword: tools/subdword_ubench.hip).  clang has
// no builtin for it; llvm_amdgcn_raw_ptr_buffer_load_format_f32 (declared above the namespace) binds the LLVM intrinsic by
// name, so that the compiler still tracks the load (s_waitcnt vmcnt) as it does for its own builtins.
// descriptor word 3 (gfx9 layout): DST_SEL_X [2:0] = 4 (X), NUM_FORMAT [14:12] = 2 (USCALED), DATA_FORMAT [18:15] = 1 (8) / 2 (16) / 4 (32)
template <typename T> __device__ __forceinline__ constexpr int sample_rsrc_flags() { return 0x00020000; }
template <> __device__ __forceinline__ constexpr int sample_rsrc_flags<unsigned char>() { return 4 | (2 << 12) | (1 << 15); }
template <> __device__ __forceinline__ constexpr int sample_rsrc_flags<unsigned short>() { return 4 | (2 << 12) | (2 << 15); }
template <typename T> __device__ __forceinline__ __amdgpu_buffer_rsrc_t sample_rsrc(const T *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(base), 0, bytes, sample_rsrc_flags<T>());
}
template <typename T> __device__ __forceinline__ float load_px(__amdgpu_buffer_rsrc_t r, int x, int row_off_bytes);
template <> __device__ __forceinline__ float load_px<float>(__amdgpu_buffer_rsrc_t r, int x, int row_off_bytes)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, x * 4, row_off_bytes, 0));
}
template <> __device__ __forceinline__ float load_px<unsigned char>(__amdgpu_buffer_rsrc_t r, int x, int row_off_bytes)
{
    return llvm_amdgcn_raw_ptr_buffer_load_format_f32(r, x, row_off_bytes, 0);
}
template <> __device__ __forceinline__ float load_px<unsigned short>(__amdgpu_buffer_rsrc_t r, int x, int row_off_bytes)
{
    return llvm_amdgcn_raw_ptr_buffer_load_format_f32(r, x * 2, row_off_bytes, 0) * 0.25f;
}
// The same sample after offset_image's -128 (combo.c:236-237, OPT_RANGE_PIXEL_OFFSET).  For 16-bit words the scaling and the
// offset go into one fused multiply-add: px * 0.25 and px * 0.25 - 128 are both exactly representable (px < 2^16), so no
// rounding happens in either form and the fused one gives the reference's bits with one instruction less.
template <typename T> __device__ __forceinline__ float load_px_offset(__amdgpu_buffer_rsrc_t r, int x, int row_off_bytes)
{
    return load_px<T>(r, x, row_off_bytes) + -128.0f;
}
template <> __device__ __forceinline__ float load_px_offset<unsigned short>(__amdgpu_buffer_rsrc_t r, int x, int row_off_bytes)
{
    return __builtin_fmaf(llvm_amdgcn_raw_ptr_buffer_load_format_f32(r, x * 2, row_off_bytes, 0), 0.25f, -128.0f);
}

// ---------------------------------------------------------------- VIF statistic
// feature/src/vif_tools.c:33-75: exponent + degree-8 polynomial in (mantissa - 1), Horner form, every step a separately
// rounded multiply and add.  Two steps of the reference's loop are dropped because they cannot change a bit:
//   the first one, 0.0f * t + c8 (t = mantissa - 1 lies in [0, 1): the product is +0, the sum c8), and
//   the last one's "+ 0.0f" (log2_poly_s[0] = 0): it could only turn a -0 product into +0, and the value goes straight
//   into e + v, which gives the same result for either zero (e is an integer-valued float or +0, never -0).
// POSITIVE: the caller guarantees x > 0 (the denominator's argument 1 + sigma1_sq / 2 >= 2), so the reference's two
// special cases (x == 0 -> -inf, x < 0 -> NaN) need no test.
template <bool POSITIVE = false> __device__ __forceinline__ float log2f_approx(float x)
{
    const uint32_t u = __float_as_uint(x);
    const float mant = __uint_as_float((u & 0x007FFFFFu) | 0x3F800000u);
    const float e = (float)((int)((u & 0x7F800000u) >> 23) - 127);
    const float t = mant - 1.0f;
    float v = -0.012671635276421f;
    v = v * t + 0.064841182402670f;
    v = v * t + -0.157048836463065f;
    v = v * t + 0.257167726303123f;
    v = v * t + -0.353800560300520f;
    v = v * t + 0.480131410397451f;
    v = v * t + -0.721314327952201f;
    v = v * t + 1.442694803896991f;
    v = v * t;
    float r = e + v;
    if (!POSITIVE) {
        r = (x == 0.0f) ? -INFINITY : r;
        r = (x < 0.0f) ? __uint_as_float(0x7FC00000u) : r;
    }
    return r;
}

// feature/src/vif_tools.c:243-263
__device__ __forceinline__ void vif_pixel(float mu1, float mu2, float xxf, float yyf, float xyf, float &num, float &den)
{
    const float sigma_nsq = 2.0f;
    const float sigma_max_inv = (float)(4.0 / (255.0 * 255.0));
    const float m1sq = mu1 * mu1, m2sq = mu2 * mu2, m12 = mu1 * mu2;
    const float s1 = xxf - m1sq;
    const float s2 = yyf - m2sq;
    const float s12 = xyf - m12;
    if (s1 < sigma_nsq) {
        // the reference evaluates 1.0 - p in double and rounds to float (vif_tools.c:248).  A single fp32 subtraction gives
        // the same bits: for |p| >= 2^-29 the double difference is exact, so both are the correctly rounded 1 - p; for
        // smaller |p| both round to 1.0f (the float neighbours of 1 are 2^-24 / 2^-23 away)
        const float p = s2 * sigma_max_inv;
        num = 1.0f - p;
        den = 1.0f;
    } else {
        const float sv = (s2 + sigma_nsq) * s1;
        const float g = sv - s12 * s12;
        const float l = log2f_approx(sv / g);
        num = (s12 < 0.0f) ? 0.0f : l;
        den = log2f_approx<true>(1.0f + s1 / sigma_nsq);           // s1 >= 2 here: the argument is >= 2
    }
}

// Tile 48 x 32 with 256 threads = four waves of 48 x 8: a wave's vertical pass has 48 + 2*8 = 64 columns at 17 taps
// (one per lane) and its horizontal pass 8 rows * (48/6) strips = 64 items; 48 divides every level of the 1080p / 4K
// pyramids (1920, 960, 480, 240).
constexpr int VIF_TW = 48, VIF_TH = 32, VIF_THREADS = 256, VIF_SV = 8, VIF_SH = 6;
constexpr int VIF_WAVES_PER_SIMD = 4;                         // HIP: 2nd __launch_bounds__ argument = min waves per SIMD;
                                                              // 35 KB LDS and <= 128 VGPRs per workgroup -> 4 workgroups per CU

// 1-D filter over a register window: out[o] = sum_k f[k] * v[o + k], in the scalar or the vector order
template <int FW, int NOUT, int ORDER>
__device__ __forceinline__ void taps_window(const float *v, float *out)
{
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
        out[o] = (ORDER == ORDER_SCALAR) ? taps_seq_at<FW, false>(v + o, 1) : (ORDER == ORDER_FMA) ? taps_fma_at<FW, false>(v + o, 1) : taps_avx_at<FW, false>(v + o, 1);
}

// Same with the order chosen per output (bit o of avx_mask): the tiles on the border frame of ORDER_AVX
// (most waves of those tiles still agree on one order for the whole window and keep its instruction-level parallelism)
template <int FW, int NOUT> __device__ __forceinline__ void taps_window_sel(const float *v, float *out, unsigned avx_mask)
{
    constexpr unsigned ALL = (1u << NOUT) - 1u;
    if (__all(avx_mask == ALL)) {
        taps_window<FW, NOUT, ORDER_AVX>(v, out);
    } else if (__all(avx_mask == 0u)) {
        taps_window<FW, NOUT, ORDER_SCALAR>(v, out);
    } else {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) out[o] = taps_sel<FW, false, ORDER_AVX>(v + o, 1, (avx_mask >> o) & 1u);
    }
}

// Tiles of a VIF scale that contain a pixel which is not avx_pure_pixel (the "border frame" of ORDER_AVX):
// tile rows {0} u [ty_b, nty) completely, and in the other rows tile columns {0} u [tx_r, ntx).
struct BorderGeom { int ntx, nty, ty_b, tx_r, nrs, ncs, ntiles; };
__host__ __device__ inline BorderGeom border_geom(int w, int h, int r, int tw, int th)
{
    BorderGeom g;
    g.ntx = (w + tw - 1) / tw;
    g.nty = (h + th - 1) / th;
    const int hb = h - r > 0 ? h - r : 0, he = avx_hcol_end(w, r) > 0 ? avx_hcol_end(w, r) : 0;
    g.ty_b = hb / th > 1 ? hb / th : 1;
    g.tx_r = he / tw > 1 ? he / tw : 1;
    if (g.ty_b > g.nty) g.ty_b = g.nty;
    if (g.tx_r > g.ntx) g.tx_r = g.ntx;
    g.nrs = 1 + g.nty - g.ty_b;
    g.ncs = 1 + g.ntx - g.tx_r;
    g.ntiles = g.nrs * g.ntx + (g.ty_b - 1) * g.ncs;
    return g;
}
__device__ __forceinline__ void border_tile_at(const BorderGeom &g, int b, int &tx, int &ty)
{
    if (b < g.nrs * g.ntx) {
        const int rr = b / g.ntx;
        tx = b - rr * g.ntx;
        ty = rr == 0 ? 0 : g.ty_b + rr - 1;
    } else {
        const int bb = b - g.nrs * g.ntx, c = bb % g.ncs;
        ty = 1 + bb / g.ncs;
        tx = c == 0 ? 0 : g.tx_r + c - 1;
    }
}

// T = sample type of the input planes (strides in samples): float for the pyramid levels and fp32 pictures, unsigned char /
// unsigned short for raw luma at scale 0 (widened in the loads, load_px)
#define VIF_KERNEL_PARAMS const T *__restrict__ ref, const T *__restrict__ dis, size_t frame_stride, int row_stride, int w, int h, \
                          double *__restrict__ partial, float *__restrict__ oref, float *__restrict__ odis,                  \
                          size_t out_frame_stride, int out_row_stride, float *__restrict__ nd
// nd (null = off): the per-pixel num and den of vif_statistic as planes, [frame][num, den][h][w] - the reference's own
// intermediate (vif.c:258-262), kept only when the caller wants the plane sums in vif_sum_s's order (vif_rowsum_kernel)
#define VIF_KERNEL_ARGS ref, dis, frame_stride, row_stride, w, h, partial, oref, odis, out_frame_stride, out_row_stride, nd

// ---------------------------------------------------------------- VIF
// Fused VIF scale. Replaces, for one scale, vif.c:189-269: two mu filters, xx/yy/xy products, three filtered moments,
// vif_statistic and vif_sum - none of the 11 intermediate planes touches HBM.  From the same input rows it also produces
// the next scale's input (vif.c:157-186): blur with the NEXT scale's filter, evaluated at even rows/columns only, so
// ref/dis are read from HBM once per scale.
// ORDER_AVX has two tile sets: BORDER = false for the tiles whose every sample is in the vector order, BORDER = true
// for the border-frame tiles, where every sample picks its order exactly as convolution_f32_avx_s_1d does
// (convolution_avx.c:806-836); both write partial[frame][tile] and the decimated planes of their own tiles.
// The four waves of a workgroup never wait for each other: every wave owns 8 output rows x 48 columns of the workgroup's
// 48x32 tile from the first load to the last sum.
//   vertical pass     lane = one of the 64 columns x0-R .. (needs 48+2R <= 64), its 8+2R input rows of ref and dis straight
//                     from global memory (buffer loads: frame descriptor + wave-uniform row offset in SGPRs, mirrored
//                     rows/columns by index), no staging tile; five quantities x 8 rows in registers
//   transpose         through the wave's own 8 x 68-float LDS rows, three planes at a time; only wave-level ordering
//                     (LDS serves one wave's instructions in order) - no __syncthreads until the final reduction
//   horizontal pass   lane = (row, strip of 6 columns); row pitch 68 = 4 mod 32: the 16 lanes (8 rows x 2 strips) that a
//                     ds_read_b64 serves per cycle start at 16 different even banks (4*row, 4*row + 6)
//   decimation        the next scale's planes for the wave's 4 even rows, same two-step scheme
// Waves whose rows lie below the picture (last tile row) skip everything but the reduction.
constexpr int VWV_TH = 8;                                    // output rows per wave
constexpr int VWV_VS = 68;                                   // LDS row pitch of the vertical results: 4 mod 32 (see above)
static_assert(VWV_TH == VIF_SV && VWV_TH * (VIF_THREADS / 64) == VIF_TH, "four waves of 8 rows make the 48x32 tile");

template <int FW> struct VifWaveGeom {
    static constexpr int R = FW / 2;
    static constexpr int VW = VIF_TW + 2 * R;                 // columns the vertical pass produces (<= 64 lanes)
    static constexpr int NV = VIF_SH + 2 * R, NV2 = NV / 2;
    static constexpr int DFW = FW == 17 ? 9 : FW == 9 ? 5 : FW == 5 ? 3 : 0;
    static constexpr int RD = DFW / 2;
    static constexpr int DVW = VIF_TW + 2 * RD;
    static constexpr int DSH = 3;
    static constexpr int VQ = 3;
    static constexpr int V_FLOATS = VQ * VWV_TH * VWV_VS;
    static constexpr int D_FLOATS = DFW ? 2 * (VWV_TH / 2) * DVW : 0;
    static constexpr int WAVE_FLOATS = (V_FLOATS + D_FLOATS + 3) / 4 * 4;
    static_assert(VW <= 64 && VW <= VWV_VS && NV % 2 == 0, "one lane per column, float2 reads");
    static_assert(VWV_TH * (VIF_TW / VIF_SH) == 64, "one horizontal item per lane");
    static_assert(2 * (VWV_TH / 2) * (VIF_TW / 2 / DSH) == 64, "one decimation strip per lane");
};

// orders the LDS traffic of ONE wave: its earlier writes are visible to its later reads (the LDS executes a wave's
// instructions in order; this only keeps the compiler from moving them across)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int FW, bool OFFSET, int ORDER, bool BORDER, typename T>
__device__ __forceinline__ void vif_wave_tile(float *smem, double *red, int tile_id, const T *__restrict__ ref, const T *__restrict__ dis,
                                              size_t frame_stride, int row_stride, int w, int h,
                                              double *__restrict__ partial,
                                              float *__restrict__ oref, float *__restrict__ odis,
                                              size_t out_frame_stride, int out_row_stride, float *__restrict__ nd)
{
    using G = VifWaveGeom<FW>;
    constexpr int R = G::R, VS = VWV_VS, DFW = G::DFW, RD = G::RD, DVW = G::DVW;
    constexpr int NW = VWV_TH + 2 * R;                        // input rows of one wave
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *vout = smem + wv * G::WAVE_FLOATS;
    float *dv_s = vout + G::V_FLOATS;                         // decimation: vertical results, [2][4][DVW]
    const int ntx = (w + VIF_TW - 1) / VIF_TW;
    const int ntiles = ntx * ((h + VIF_TH - 1) / VIF_TH);
    const int x0 = (tile_id % ntx) * VIF_TW;
    const int y0 = (tile_id / ntx) * VIF_TH + wv * VWV_TH;    // this wave's first output row (wave-uniform)
    double acc[2] = { 0.0, 0.0 };

    if (y0 < h) {
        const T *fr = ref + (size_t)blockIdx.z * frame_stride;
        const T *fd = dis + (size_t)blockIdx.z * frame_stride;
        // ---- vertical pass: lane = column x0 - R + lane
        float vo[5][VWV_TH];
        const int vc = lane;
        const int gxc = mirror_idx(x0 - R + vc, w);
        {
            float rv[NW], dv[NW], t[NW];
            const unsigned frame_bytes = (unsigned)h * (unsigned)row_stride * (unsigned)sizeof(T);
            const auto br = sample_rsrc(fr, frame_bytes), bd = sample_rsrc(fd, frame_bytes);
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int roff = mirror_idx(y0 - R + i, h) * row_stride * (int)sizeof(T);   // wave-uniform: SGPR
                rv[i] = OFFSET ? load_px_offset<T>(br, gxc, roff) : load_px<T>(br, gxc, roff);       // offset_image, combo.c:236-237
                dv[i] = OFFSET ? load_px_offset<T>(bd, gxc, roff) : load_px<T>(bd, gxc, roff);
            }
            if (DFW > 0) {
                // decimation, vertical pass at the even rows (y0 is even): output row y0 + 2i is window element R + 2i
                const bool dcol = vc >= R - RD && vc < R - RD + DVW;
                const bool col_avx = ORDER == ORDER_AVX && avx_vcol_ok(gxc, w);
#pragma unroll
                for (int i = 0; i < VWV_TH / 2; ++i) {
                    const bool avx = col_avx && avx_row_ok(y0 + 2 * i, h, RD);
                    const float a = taps_sel<DFW ? DFW : 3, false, ORDER>(rv + R - RD + 2 * i, 1, avx);
                    const float b = taps_sel<DFW ? DFW : 3, false, ORDER>(dv + R - RD + 2 * i, 1, avx);
                    if (dcol) {
                        dv_s[i * DVW + vc - (R - RD)] = a;
                        dv_s[(VWV_TH / 2 + i) * DVW + vc - (R - RD)] = b;
                    }
                }
            }
            unsigned vmask = 0;                               // BORDER: vector order iff row and column both allow it
            if (BORDER && avx_vcol_ok(gxc, w)) {
#pragma unroll
                for (int o = 0; o < VWV_TH; ++o) vmask |= avx_row_ok(y0 + o, h, R) ? 1u << o : 0u;
            }
#define VIF_VPASS(src, q)                                                  \
            if (BORDER) taps_window_sel<FW, VWV_TH>(src, vo[q], vmask);    \
            else taps_window<FW, VWV_TH, ORDER>(src, vo[q])
            VIF_VPASS(rv, 0);
            VIF_VPASS(dv, 1);
#pragma unroll
            for (int i = 0; i < NW; ++i) t[i] = rv[i] * rv[i];
            VIF_VPASS(t, 2);
#pragma unroll
            for (int i = 0; i < NW; ++i) t[i] = dv[i] * dv[i];
            VIF_VPASS(t, 3);
#pragma unroll
            for (int i = 0; i < NW; ++i) t[i] = rv[i] * dv[i];
            VIF_VPASS(t, 4);
#undef VIF_VPASS
        }
        if (DFW > 0) {
            // decimation, horizontal pass at the even columns (x0 is even) -> next scale's planes;
            // lane = (plane, output row, strip of DSH outputs), taps from a register window
            wave_lds_sync();
            constexpr int DSH = G::DSH, DNW = 2 * (DSH - 1) + (DFW ? DFW : 1);
            const int pl = lane >> 5, oy = (lane >> 3) & 3, ox0 = (lane & 7) * DSH;
            const int ow = w / 2, oh = h / 2, dend = avx_hcol_end(w, RD);
            const float *src = dv_s + (pl * (VWV_TH / 2) + oy) * DVW + 2 * ox0;
            float dw[DNW];
#pragma unroll
            for (int k = 0; k < DNW; ++k) dw[k] = src[k];
            const int gy = y0 / 2 + oy;
            float *dst = (pl ? odis : oref) + (size_t)blockIdx.z * out_frame_stride + (size_t)gy * out_row_stride + x0 / 2 + ox0;
#pragma unroll
            for (int o = 0; o < DSH; ++o) {
                const int sx = x0 + 2 * (ox0 + o);
                const float val = taps_sel<DFW ? DFW : 3, false, ORDER>(dw + 2 * o, 1, ORDER == ORDER_AVX && sx >= RD && sx < dend);
                if (x0 / 2 + ox0 + o < ow && gy < oh) dst[o] = val;
            }
        }
        // ---- horizontal pass: lane = (row, strip of VIF_SH columns); planes go through LDS in rounds of VQ
        const int row = lane & 7, strip = lane >> 3;
        float ho[5][VIF_SH];
        unsigned hmask = 0;
        if (BORDER) {
            const int hend = avx_hcol_end(w, R);
#pragma unroll
            for (int o = 0; o < VIF_SH; ++o) {
                const int gx = x0 + strip * VIF_SH + o;
                hmask |= (gx >= R && gx < hend) ? 1u << o : 0u;
            }
        }
#pragma unroll
        for (int round = 0; round < (5 + G::VQ - 1) / G::VQ; ++round) {
            constexpr int VQ = G::VQ;
            const int q0 = round * VQ, nq = 5 - q0 < VQ ? 5 - q0 : VQ;
            wave_lds_sync();                                  // the previous round's reads stay before these writes
#pragma unroll
            for (int q = 0; q < VQ; ++q)
                if (q < nq) {
#pragma unroll
                    for (int o = 0; o < VWV_TH; ++o) vout[(q * VWV_TH + o) * VS + vc] = vo[q0 + q][o];
                }
            wave_lds_sync();
#pragma unroll
            for (int q = 0; q < VQ; ++q)
                if (q < nq) {
                    float v[G::NV];
                    const float2 *p = reinterpret_cast<const float2 *>(vout + (q * VWV_TH + row) * VS + strip * VIF_SH);
#pragma unroll
                    for (int m = 0; m < G::NV2; ++m) {
                        const float2 f2 = p[m];
                        v[2 * m] = f2.x; v[2 * m + 1] = f2.y;
                    }
                    if (BORDER) taps_window_sel<FW, VIF_SH>(v, ho[q0 + q], hmask);
                    else taps_window<FW, VIF_SH, ORDER>(v, ho[q0 + q]);
                }
        }
        const int gy = y0 + row;
#pragma unroll
        for (int o = 0; o < VIF_SH; ++o) {
            const int gx = x0 + strip * VIF_SH + o;
            float n, d;
            vif_pixel(ho[0][o], ho[1][o], ho[2][o], ho[3][o], ho[4][o], n, d);
            // the interior tiles of ORDER_AVX lie inside the picture; elsewhere a tile may hang over its right / bottom edge
            const bool mine = (ORDER == ORDER_AVX && !BORDER) || (gx < w && gy < h);
            if (mine) { acc[0] += (double)n; acc[1] += (double)d; }
            if (nd && mine) {                                 // (wave-uniform test first)
                float *pn = nd + ((size_t)blockIdx.z * 2 * h + gy) * w + gx;
                pn[0] = n;
                pn[(size_t)h * w] = d;
            }
        }
    }
    const size_t tile = (size_t)blockIdx.z * ntiles + tile_id;
    block_reduce_store<2, VIF_THREADS / 64>(acc, red, partial + tile * 2);
}

template <int FW, bool OFFSET, int ORDER, typename T>
__global__ __launch_bounds__(VIF_THREADS, VIF_WAVES_PER_SIMD) void vif_stat_kernel(VIF_KERNEL_PARAMS)
{
    __shared__ __attribute__((aligned(16))) float smem[(VIF_THREADS / 64) * VifWaveGeom<FW>::WAVE_FLOATS];
    __shared__ double red[2 * (VIF_THREADS / 64)];
    vif_wave_tile<FW, OFFSET, ORDER, false, T>(smem, red, xcd_tile_index(), VIF_KERNEL_ARGS);
}

template <int FW, bool OFFSET, typename T>
__global__ __launch_bounds__(VIF_THREADS, VIF_WAVES_PER_SIMD) void vif_stat_avx_kernel(VIF_KERNEL_PARAMS, int nb_pad)
{
    __shared__ __attribute__((aligned(16))) float smem[(VIF_THREADS / 64) * VifWaveGeom<FW>::WAVE_FLOATS];
    __shared__ double red[2 * (VIF_THREADS / 64)];
    const BorderGeom bg = border_geom(w, h, FW / 2, VIF_TW, VIF_TH);
    const int ntx = bg.ntx;
    const int b = blockIdx.x;
    if (b < nb_pad) {
        if (b >= bg.ntiles) return;                           // padding workgroup
        int tx, ty;
        border_tile_at(bg, b, tx, ty);
        vif_wave_tile<FW, OFFSET, ORDER_AVX, true, T>(smem, red, ty * ntx + tx, VIF_KERNEL_ARGS);
    } else {
        const int i = xcd_tile_index(b - nb_pad, (int)gridDim.x - nb_pad), iw = bg.tx_r - 1;
        vif_wave_tile<FW, OFFSET, ORDER_AVX, false, T>(smem, red, (1 + i / iw) * ntx + 1 + i % iw, VIF_KERNEL_ARGS);
    }
}

// ---------------------------------------------------------------- VIF, workgroup-wide tiles
// Same arithmetic as vif_wave_tile, other work distribution, for planes at least 240 wide whose width is a multiple of 8
// (every level of the 1080p and 4K pyramids tiles exactly; elsewhere the last tile of a row is narrower).  A wave-private 48-column tile spends a third of its vertical pass on halo columns
// (64 filtered columns per 48 outputs at 17 taps); here the four waves of a workgroup stand side by side on 256 columns
// and share the vertical results through LDS: 256 filtered columns per 240 outputs.
//   vertical pass     thread = one of the 256 columns x0-R .. ; its 8+2R input rows from global memory, {r, d, r^2, d^2, rd}
//                     x 8 rows in registers; the vertical half of the next scale's decimation from the same windows
//   horizontal pass   thread = (row, strip of 8 columns): 8 x 30 = 240 of the 256 threads; planes through LDS in rounds of 3,
//                     row pitch 258 = 2 mod 32 and a thread order that puts 4 rows x 4 strips into every group of 16 lanes:
//                     the 16 lanes a ds_read_b64 serves per cycle touch 32 different banks
//   ORDER_AVX         rows pick their order per output (uniform over the workgroup); horizontally only the first and the
//                     last 8 columns of a plane ever leave the vector order (w % 8 == 0 here): those two EDGE STRIPS are
//                     redone by one wave, one pixel per lane, in the order convolution_f32_avx_s_1d takes per sample,
//                     instead of making a whole wave evaluate both orders for every output
constexpr int VWG_TW = 240, VWG_TH = 8, VWG_COLS = 256, VWG_P = 258, VWG_HS = 8, VWG_STRIPS = VWG_TW / VWG_HS;
static_assert(VWG_COLS == VIF_THREADS && VWG_TW + 16 == VWG_COLS && VWG_P % 32 == 2, "one thread per filtered column; pitch 2 mod 32");

template <int FW> struct VifWgGeom {
    static constexpr int R = FW / 2;
    static constexpr int NV = VWG_HS + 2 * R;                  // window of one horizontal item
    static constexpr int DFW = FW == 17 ? 9 : FW == 9 ? 5 : FW == 5 ? 3 : 0;
    static constexpr int RD = DFW / 2;
    static constexpr int DVW = VWG_TW + 2 * RD;                // columns the decimation's vertical pass keeps
    static constexpr int DP = 250;                             // its row pitch
    static constexpr int DSH = 4;                              // decimated outputs per thread: 2 planes x 4 rows x 30 strips = 240 threads
    static constexpr int VQ = 3;
    static constexpr int V_FLOATS = VQ * VWG_TH * VWG_P;
    static constexpr int D_FLOATS = DFW ? 2 * (VWG_TH / 2) * DP : 0;
    static_assert(NV % 2 == 0 && DVW <= DP && VWG_TW + 2 * R <= VWG_COLS, "float2 reads; planes fit");
};

// horizontal item of a thread: 4 rows x 4 strips per 16 lanes (see above); threads 224..239 take the last two strips
__device__ __forceinline__ void vwg_item(int tid, int &row, int &strip)
{
    if (tid < 224) {
        const int g = tid >> 4, in = tid & 15;
        strip = (g >> 1) * 4 + (in & 3);
        row = (g & 1) * 4 + (in >> 2);
    } else {
        strip = 28 + (tid & 1);
        row = (tid >> 1) & 7;
    }
}

template <int FW, bool OFFSET, int ORDER, typename T>
__global__ __launch_bounds__(VIF_THREADS, VIF_WAVES_PER_SIMD) void vif_stat_wg_kernel(VIF_KERNEL_PARAMS)
{
    using G = VifWgGeom<FW>;
    constexpr int R = G::R, P = VWG_P, DFW = G::DFW, RD = G::RD, DP = G::DP;
    constexpr int NW = VWG_TH + 2 * R;
    __shared__ __attribute__((aligned(16))) float smem[G::V_FLOATS + G::D_FLOATS];
    __shared__ double red[2 * (VIF_THREADS / 64)];
    float *vout = smem, *dv_s = smem + G::V_FLOATS;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntx = (w + VWG_TW - 1) / VWG_TW;
    const int ntiles = gridDim.x;
    const int tile_id = xcd_tile_index();
    const int tx = tile_id % ntx;
    const int x0 = tx * VWG_TW, y0 = (tile_id / ntx) * VWG_TH;
    // the last tile of a row may be narrower (w % 8 == 0: a whole number of strips); waves and threads beyond it idle
    const int tw = min(VWG_TW, w - x0), nstrips = tw / VWG_HS;
    const T *fr = ref + (size_t)blockIdx.z * frame_stride;
    const T *fd = dis + (size_t)blockIdx.z * frame_stride;
    double acc[2] = { 0.0, 0.0 };

    // ---- vertical pass: thread = column x0 - R + tid
    float vo[5][VWG_TH];
    const int vc = tid;
    const bool vlane = vc < tw + 2 * R;                        // FW < 17 or a narrow last tile: fewer than 256 columns
    const int gxc = mirror_idx(x0 - R + vc, w);
    if (wv * 64 < tw + 2 * R) {                                // wave-uniform: a wave without columns skips the pass
        float rv[NW], dv[NW], t[NW];
        const unsigned frame_bytes = (unsigned)h * (unsigned)row_stride * (unsigned)sizeof(T);
        const auto br = sample_rsrc(fr, frame_bytes), bd = sample_rsrc(fd, frame_bytes);
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int roff = mirror_idx(y0 - R + i, h) * row_stride * (int)sizeof(T);   // workgroup-uniform: SGPR
            rv[i] = OFFSET ? load_px_offset<T>(br, gxc, roff) : load_px<T>(br, gxc, roff);           // offset_image, combo.c:236-237
            dv[i] = OFFSET ? load_px_offset<T>(bd, gxc, roff) : load_px<T>(bd, gxc, roff);
        }
        if (DFW > 0) {
            // decimation, vertical pass at the even rows (y0 is even): output row y0 + 2i is window element R + 2i.
            // w % 8 == 0: every column is in the vector order of the vertical pass (convolution_avx.c:806-815)
            const bool dcol = vc >= R - RD && vc < R - RD + tw + 2 * RD;
#pragma unroll
            for (int i = 0; i < VWG_TH / 2; ++i) {
                const bool avx = ORDER == ORDER_AVX && avx_row_ok(y0 + 2 * i, h, RD);
                const float a = taps_sel<DFW ? DFW : 3, false, ORDER>(rv + R - RD + 2 * i, 1, avx);
                const float b = taps_sel<DFW ? DFW : 3, false, ORDER>(dv + R - RD + 2 * i, 1, avx);
                if (dcol) {
                    dv_s[i * DP + vc - (R - RD)] = a;
                    dv_s[(VWG_TH / 2 + i) * DP + vc - (R - RD)] = b;
                }
            }
        }
        unsigned vmask = 0;                                    // ORDER_AVX: vector order in the rows r <= y < h - r
        if (ORDER == ORDER_AVX) {
#pragma unroll
            for (int o = 0; o < VWG_TH; ++o) vmask |= avx_row_ok(y0 + o, h, R) ? 1u << o : 0u;
        }
#define VWG_VPASS(src, q)                                                          \
        if (ORDER == ORDER_AVX) taps_window_sel<FW, VWG_TH>(src, vo[q], vmask);    \
        else taps_window<FW, VWG_TH, ORDER>(src, vo[q])
        VWG_VPASS(rv, 0);
        VWG_VPASS(dv, 1);
#pragma unroll
        for (int i = 0; i < NW; ++i) t[i] = rv[i] * rv[i];
        VWG_VPASS(t, 2);
#pragma unroll
        for (int i = 0; i < NW; ++i) t[i] = dv[i] * dv[i];
        VWG_VPASS(t, 3);
#pragma unroll
        for (int i = 0; i < NW; ++i) t[i] = rv[i] * dv[i];
        VWG_VPASS(t, 4);
#undef VWG_VPASS
    }

    // ---- horizontal pass
    int row, strip;
    vwg_item(tid, row, strip);
    const bool hlane = tid < VWG_TH * VWG_STRIPS && strip < nstrips;
    // edge strips (ORDER_AVX only): columns [0, 8) of the plane and [w - 8, w); one wave each, lane = (row, column)
    const int hend = avx_hcol_end(w, R);
    const bool edge_l = ORDER == ORDER_AVX && tx == 0, edge_r = ORDER == ORDER_AVX && tx == ntx - 1;
    const bool edge_wave = (edge_l && wv == 0) || (edge_r && wv == 1);          // wave-uniform
    const int lane = tid & 63;
    const int erow = lane & 7, ecol = (wv == 0 ? 0 : tw - VWG_HS) + (lane >> 3);       // column inside the tile
    float ho[5][VWG_HS], he[5];
#pragma unroll
    for (int round = 0; round < (5 + G::VQ - 1) / G::VQ; ++round) {
        constexpr int VQ = G::VQ;
        const int q0 = round * VQ, nq = 5 - q0 < VQ ? 5 - q0 : VQ;
        if (round > 0) __syncthreads();                        // the previous round's reads are done
        if (vlane) {
#pragma unroll
            for (int q = 0; q < VQ; ++q)
                if (q < nq) {
#pragma unroll
                    for (int o = 0; o < VWG_TH; ++o) vout[(q * VWG_TH + o) * P + vc] = vo[q0 + q][o];
                }
        }
        __syncthreads();
        if (round == 0 && DFW > 0) {
            // decimation, horizontal pass at the even columns (x0 is even) -> next scale's planes; thread = (plane, output
            // row, strip of 4 outputs): a row of a tile leaves as 120 consecutive floats, one 16-byte store per thread
            constexpr int DSH = G::DSH, DNW = 2 * (DSH - 1) + (DFW ? DFW : 1);
            const int ds = tid % (VWG_TW / 2 / DSH), dr = (tid / (VWG_TW / 2 / DSH)) & 3, pl = tid / (4 * (VWG_TW / 2 / DSH));
            const int ox0 = ds * DSH;
            if (tid < 2 * (VWG_TH / 2) * (VWG_TW / 2 / DSH) && 2 * ox0 < tw) {
                const int oh = h / 2, dend = avx_hcol_end(w, RD);
                const float *src = dv_s + (pl * (VWG_TH / 2) + dr) * DP + 2 * ox0;
                float dw[DNW], val[DSH];
#pragma unroll
                for (int k = 0; k < DNW; ++k) dw[k] = src[k];
#pragma unroll
                for (int o = 0; o < DSH; ++o) {
                    const int sx = x0 + 2 * (ox0 + o);
                    val[o] = taps_sel<DFW ? DFW : 3, false, ORDER>(dw + 2 * o, 1, ORDER == ORDER_AVX && sx >= RD && sx < dend);
                }
                const int gyd = y0 / 2 + dr;
                if (gyd < oh) {
                    float *dst = (pl ? odis : oref) + (size_t)blockIdx.z * out_frame_stride + (size_t)gyd * out_row_stride + x0 / 2 + ox0;
                    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
                        *reinterpret_cast<float4 *>(dst) = make_float4(val[0], val[1], val[2], val[3]);
                    } else {
#pragma unroll
                        for (int o = 0; o < DSH; ++o) dst[o] = val[o];
                    }
                }
            }
        }
        if (hlane) {
#pragma unroll
            for (int q = 0; q < VQ; ++q)
                if (q < nq) {
                    float v[G::NV];
                    const float2 *p = reinterpret_cast<const float2 *>(vout + (q * VWG_TH + row) * P + strip * VWG_HS);
#pragma unroll
                    for (int m = 0; m < G::NV / 2; ++m) {
                        const float2 f2 = p[m];
                        v[2 * m] = f2.x; v[2 * m + 1] = f2.y;
                    }
                    taps_window<FW, VWG_HS, ORDER>(v, ho[q0 + q]);
                }
        }
        if (edge_wave) {
            const int gx = x0 + ecol;
            const bool avx = gx >= R && gx < hend;
#pragma unroll
            for (int q = 0; q < VQ; ++q)
                if (q < nq) {
                    float v[FW];
                    const float *p = vout + (q * VWG_TH + erow) * P + ecol;
#pragma unroll
                    for (int k = 0; k < FW; ++k) v[k] = p[k];
                    he[q0 + q] = taps_sel<FW, false, ORDER_AVX>(v, 1, avx);
                }
        }
    }
    if (hlane) {
        const int gy = y0 + row;
        const bool in_edge = (edge_l && strip == 0) || (edge_r && strip == nstrips - 1);
        if (gy < h && !in_edge) {
#pragma unroll
            for (int o = 0; o < VWG_HS; ++o) {
                float n, d;
                vif_pixel(ho[0][o], ho[1][o], ho[2][o], ho[3][o], ho[4][o], n, d);
                acc[0] += (double)n; acc[1] += (double)d;
                if (nd) {
                    float *pn = nd + ((size_t)blockIdx.z * 2 * h + gy) * w + x0 + strip * VWG_HS + o;
                    pn[0] = n;
                    pn[(size_t)h * w] = d;
                }
            }
        }
    }
    if (edge_wave && y0 + erow < h) {
        float n, d;
        vif_pixel(he[0], he[1], he[2], he[3], he[4], n, d);
        acc[0] += (double)n; acc[1] += (double)d;
        if (nd) {
            float *pn = nd + ((size_t)blockIdx.z * 2 * h + y0 + erow) * w + x0 + ecol;
            pn[0] = n;
            pn[(size_t)h * w] = d;
        }
    }
    const size_t tile = (size_t)blockIdx.z * ntiles + tile_id;
    block_reduce_store<2, VIF_THREADS / 64>(acc, red, partial + tile * 2);
}

// vif_sum_s (vif_tools.c:161-179) in its own order, for callers that want the plane sums to the reference's last bit: one
// fp32 accumulator per row, columns left to right, then the row sums top to bottom in fp32.  The VIF kernels leave the
// per-pixel num and den as planes (nd); lane = one row walks it 4 samples per load, 64 rows per wave side by side.
__global__ __launch_bounds__(64) void vif_rowsum_kernel(const float *__restrict__ nd, int w, int h, float *__restrict__ rowsum)
{
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= h) return;
    const size_t plane = (size_t)blockIdx.z * 2 + blockIdx.y;         // [frame][num, den]
    const float *p = nd + (plane * h + row) * w;
    float acc = 0.0f;
    int j = 0;
    if ((w & 3) == 0) {
        // eight loads in flight before the first of their 32 additions: the chain waits for memory once per 32 samples
        for (; j + 32 <= w; j += 32) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4 *>(p + j + 4 * k);
#pragma unroll
            for (int k = 0; k < 8; ++k) { acc = acc + v[k].x; acc = acc + v[k].y; acc = acc + v[k].z; acc = acc + v[k].w; }
        }
        for (; j < w; j += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(p + j);
            acc = acc + v.x; acc = acc + v.y; acc = acc + v.z; acc = acc + v.w;
        }
    }
    for (; j < w; ++j) acc = acc + p[j];
    rowsum[plane * h + row] = acc;
}
// accum += accum_inner over the rows; one wave per (frame, num / den); the total goes to out[frame * out_stride + 2 * scale + plane]
__global__ __launch_bounds__(64) void vif_total_kernel(const float *__restrict__ rowsum, int h, double *__restrict__ out, size_t out_stride, int scale)
{
    const int lane = threadIdx.x;
    const size_t plane = (size_t)blockIdx.y * 2 + blockIdx.x;
    const float *rs = rowsum + plane * h;
    float acc = 0.0f;
    for (int r0 = 0; r0 < h; r0 += 64) {
        const float v = r0 + lane < h ? rs[r0 + lane] : 0.0f;        // + 0.0f beyond the last row: no change
#pragma unroll
        for (int i = 0; i < 64; ++i) acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i));
    }
    if (lane == 0) out[(size_t)blockIdx.y * out_stride + 2 * scale + blockIdx.x] = (double)acc;
}

// ---------------------------------------------------------------- ADM
// One workgroup = ADM_SW x ADM_SH coefficient slots (a ADM_TW x ADM_TH interior plus the one-coefficient halo the
// masking threshold looks at), 256 threads, no staging of the input picture:
//   phase 1  vertical db2 pass straight from global memory: thread = one input column x half of the slot rows,
//            its 18 input rows are all in flight before the first use; the four vertical sums of a (slot row,
//            column) go to LDS as one float4, columns split into an even and an odd plane so that
//   phase 2  the horizontal pass of slot tx reads float4s [tx] and [tx+1] of both planes (conflict free);
//            thread = slot column x 4 slot rows; decouple + CSF stay in registers, (1/30)|csf_a| goes to LDS
//   phase 3  threshold from the 3x3 neighbourhood, contrast masking, cube sums.
constexpr int ADM_THREADS = 256;
constexpr int ADM_SW = 63, ADM_SH = 16;                        // slots: 2*63+2 = 128 input columns = two waves of lanes
constexpr int ADM_TW = ADM_SW - 2, ADM_TH = ADM_SH - 2;        // interior coefficients per workgroup
constexpr int ADM_COLS = 2 * ADM_SW + 2;                       // 128
constexpr int ADM_RPT = ADM_SH / 2;                            // slot rows per thread in phase 1 (8)
constexpr int ADM_SPT = ADM_SH / 4;                            // slots per thread in phases 2 and 3 (4)
constexpr int ADM_VP = 72;                                     // float4s per (slot row, parity) plane: 64 + 8 so that the two
                                                               // parities of a 16-lane group fall into different banks
constexpr int ADM_AW = 64;                                     // row pitch of the |csf_a| planes
static_assert(ADM_COLS == 128 && ADM_THREADS == 2 * ADM_COLS, "phase 1 maps thread -> (column, row half)");

// One ADM scale, fused (adm.c:132-247 for one value of `scale`): adm_dwt2 x2, adm_decouple, adm_csf x3,
// adm_cm_thresh, adm_cm and the six adm_sum_cube accumulations.  Only the LL bands go back to HBM.
// FAST (ORDER_FMA, opt-in): the 4-tap db2 sums with fused multiply-adds; everything after the DWT is unchanged.
template <bool FAST> __device__ __forceinline__ float db2_acc(float acc, float c, float x) { return FAST ? __builtin_fmaf(c, x, acc) : acc + c * x; }

template <bool OFFSET, bool FAST, typename T>
__global__ __launch_bounds__(ADM_THREADS, 4) void adm_scale_kernel(const T *__restrict__ ref, const T *__restrict__ dis,
                                                                   size_t in_frame_stride, int in_row_stride, int w, int h,
                                                                   float *__restrict__ ll_ref, float *__restrict__ ll_dis,
                                                                   size_t ll_frame_stride, int ll_row_stride,
                                                                   float rfac_hv, float rfac_d,
                                                                   int left, int top, int right, int bottom,
                                                                   double *__restrict__ partial)
{
    __shared__ float4 vbuf[ADM_SH * 2 * ADM_VP];               // [slot row][column parity][column / 2] = {lo_r, hi_r, lo_d, hi_d}
    float *absa = reinterpret_cast<float *>(vbuf);             // after phase 2: (1/30)|csf_a|, [band h,v,d][ADM_SH][ADM_AW]
    double *red = reinterpret_cast<double *>(vbuf);

    const int tid = threadIdx.x;
    const int bw = (w + 1) / 2, bh = (h + 1) / 2;
    const int ntx = (bw + ADM_TW - 1) / ADM_TW;
    // XCD-aware order: the workgroups one XCD receives (round-robin dispatch: blocks b, b+8, ..) cover a contiguous run of
    // tiles = a band of tile rows, so that the halo rows of vertically neighbouring tiles meet in that XCD's L2.  The rows
    // above and below the centre window are cheap (see in_window below), which would leave the XCDs with the outer bands
    // idle at the end of the launch - so the band an XCD takes rotates from frame to frame (dealing the tile rows round-robin
    // over the XCDs inside every frame balances too, but vertical neighbours then never share an L2: input read 1.21x)
    int tile_id;
    {
        const int b = blockIdx.x, nt = gridDim.x;
        const int xcd = (b + blockIdx.z) & 7;                  // the run this workgroup's XCD takes in this frame
        const int q = nt >> 3, r = nt & 7;
        // runs differ in length by at most one tile; a rotated XCD may hold one block more or less than its run: the
        // index is taken modulo the run layout, which stays a bijection because (b >> 3) enumerates every run's blocks
        const int run_len = q + (xcd < r ? 1 : 0), run_start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int j = b >> 3;
        tile_id = j < run_len ? run_start + j : -1;
        if (tile_id < 0) {
            // the block of a longer physical slot that has no tile in the (shorter) rotated run: it takes the tile the
            // opposite rotation leaves over - the last tile of the run that is longer than its physical slot
            const int src = b & 7;                             // physical slot: has q + 1 blocks, so src < r
            // runs with q + 1 tiles whose rotated slot has only q blocks: rotated run index x with x < r and ((x - z) & 7) >= r
            int k = 0, want = 0;
            for (int s2 = 0; s2 < src; ++s2) want += (((s2 + blockIdx.z) & 7) >= r) ? 1 : 0;   // how many such blocks precede this one
            for (int x = 0; x < 8; ++x) {
                if (x < r && ((x - (int)blockIdx.z) & 7) >= r) {
                    if (k == want) { tile_id = x * (q + 1) + q; }
                    ++k;
                }
            }
        }
    }
    const int cx0 = (tile_id % ntx) * ADM_TW, cy0 = (tile_id / ntx) * ADM_TH;   // first interior coefficient
    const T *fr = ref + (size_t)blockIdx.z * in_frame_stride;
    const T *fd = dis + (size_t)blockIdx.z * in_frame_stride;
    // The six sums only look at the centre window [left, right) x [top, bottom) (adm_sum_cube, adm_tools.c:55-60), and a
    // coefficient's masked value depends on its 3x3 neighbourhood only, which every workgroup computes for itself.  A
    // workgroup whose interior coefficients all lie outside the window (28 % of them at 1080p) therefore contributes
    // nothing but its LL band: it runs the low-pass halves of the two DWT passes and skips everything else.
    const bool in_window = cx0 < right && cx0 + ADM_TW > left && cy0 < bottom && cy0 + ADM_TH > top;   // workgroup-uniform
    // A tile none of whose slots (halo included) lies on or beyond the picture's outermost coefficients, and none of whose
    // pixel taps leaves the picture, needs no mirror rule anywhere (workgroup-uniform; most tiles): its rows and columns are
    // consecutive and the halo slots of phase 3 have real neighbours.  The general path costs such a tile ~300 scalar and
    // branch instructions per wave that decide nothing.
    const bool inner_tile = cx0 >= 3 && cx0 + ADM_SW < bw - 1 && cy0 >= 3 && cy0 + ADM_SH < bh - 1 &&
                            2 * cx0 - 3 + ADM_COLS <= w && 2 * (cy0 - 1 + ADM_SH) + 2 < h;
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
    if (in_window || ll_ref) {

    // ---- phase 1: vertical pass (adm_tools.c:313-334), taps of coefficient row cy at pixel rows 2cy-1 .. 2cy+2
    {
        const int l = tid & (ADM_COLS - 1);
        const int half = __builtin_amdgcn_readfirstlane(tid >> 7);                // wave-uniform
        const int ty0 = half * ADM_RPT;
        const int iy0 = 2 * (cy0 - 1 + ty0) - 1;
        float pr[2 * ADM_RPT + 2], pd[2 * ADM_RPT + 2];
        const unsigned frame_bytes = (unsigned)h * (unsigned)in_row_stride * (unsigned)sizeof(T);
        const auto br = sample_rsrc(fr, frame_bytes), bd = sample_rsrc(fd, frame_bytes);
        if (inner_tile) {
            const int xc = 2 * cx0 - 3 + l;                                        // slot 0 = coefficient cx0-1, tap 0 = pixel 2(cx0-1)-1
            const int roff0 = iy0 * in_row_stride * (int)sizeof(T);                // wave-uniform: SGPR
#pragma unroll
            for (int k = 0; k < 2 * ADM_RPT + 2; ++k) {
                const int roff = roff0 + k * in_row_stride * (int)sizeof(T);
                pr[k] = OFFSET ? load_px_offset<T>(br, xc, roff) : load_px<T>(br, xc, roff);           // offset_image, combo.c:236-237
                pd[k] = OFFSET ? load_px_offset<T>(bd, xc, roff) : load_px<T>(bd, xc, roff);
            }
        } else {
            const int xc = mirror_idx(2 * cx0 - 3 + l, w);
#pragma unroll
            for (int k = 0; k < 2 * ADM_RPT + 2; ++k) {
                const int roff = mirror_idx(iy0 + k, h) * in_row_stride * (int)sizeof(T);
                pr[k] = OFFSET ? load_px_offset<T>(br, xc, roff) : load_px<T>(br, xc, roff);
                pd[k] = OFFSET ? load_px_offset<T>(bd, xc, roff) : load_px<T>(bd, xc, roff);
            }
        }
        float4 *vdst = vbuf + (ty0 * 2 + (l & 1)) * ADM_VP + (l >> 1);
#pragma unroll
        for (int i = 0; i < ADM_RPT; ++i) {
            // sums start from the first product: 0.0f + x == x except for the sign of a zero, which nothing downstream sees
            float lo_r = kDb2Lo[0] * pr[2 * i], lo_d = kDb2Lo[0] * pd[2 * i], hi_r = 0.0f, hi_d = 0.0f;
#pragma unroll
            for (int t = 1; t < 4; ++t) {
                lo_r = db2_acc<FAST>(lo_r, kDb2Lo[t], pr[2 * i + t]);
                lo_d = db2_acc<FAST>(lo_d, kDb2Lo[t], pd[2 * i + t]);
            }
            if (in_window) {
                hi_r = kDb2Hi[0] * pr[2 * i];
                hi_d = kDb2Hi[0] * pd[2 * i];
#pragma unroll
                for (int t = 1; t < 4; ++t) {
                    hi_r = db2_acc<FAST>(hi_r, kDb2Hi[t], pr[2 * i + t]);
                    hi_d = db2_acc<FAST>(hi_d, kDb2Hi[t], pd[2 * i + t]);
                }
            }
            vdst[i * 2 * ADM_VP] = make_float4(lo_r, hi_r, lo_d, hi_d);
        }
    }
    __syncthreads();

    // ---- phase 2: horizontal pass + decouple + CSF; thread = slot column tx, slot rows 4*wave .. 4*wave+3
    const int tx = tid & 63, wv = tid >> 6;
    const bool lane_on = tx < ADM_SW;
    const int cxr = cx0 - 1 + tx;
    if (!in_window) {
        // outside the window: band_a of ref and dis only (adm_tools.c:336-361, the low-pass taps on the low-pass rows)
#pragma unroll
        for (int s = 0; s < ADM_SPT; ++s) {
            const int ty = wv * ADM_SPT + s;
            const int cyr = cy0 - 1 + ty;
            const int txc = lane_on ? tx : 0;
            const float4 *row = vbuf + ty * 2 * ADM_VP;
            const float4 q[4] = { row[txc], row[ADM_VP + txc], row[txc + 1], row[ADM_VP + txc + 1] };
            float ra = kDb2Lo[0] * q[0].x, da = kDb2Lo[0] * q[0].z;
#pragma unroll
            for (int t = 1; t < 4; ++t) {
                ra = db2_acc<FAST>(ra, kDb2Lo[t], q[t].x);
                da = db2_acc<FAST>(da, kDb2Lo[t], q[t].z);
            }
            if (lane_on && tx >= 1 && tx <= ADM_TW && ty >= 1 && ty <= ADM_TH && cxr < bw && cyr < bh) {
                const size_t g = (size_t)blockIdx.z * ll_frame_stride + (size_t)cyr * ll_row_stride + cxr;
                ll_ref[g] = ra;
                ll_dis[g] = da;
            }
        }
    } else {
    float keep_r[ADM_SPT][3], keep_o[ADM_SPT][3], p30[ADM_SPT][3], p15[ADM_SPT][3];
    const float cos1sq = (float)(0.99984769515639127 * 0.99984769515639127);   // cos(1 deg)^2, adm_tools.c:82
#pragma unroll
    for (int s = 0; s < ADM_SPT; ++s) {
        const int ty = wv * ADM_SPT + s;
        const int cyr = cy0 - 1 + ty;
        const int txc = lane_on ? tx : 0;
        const float4 *row = vbuf + ty * 2 * ADM_VP;
        const float4 q[4] = { row[txc], row[ADM_VP + txc], row[txc + 1], row[ADM_VP + txc + 1] };   // pixel columns 2cx-1 .. 2cx+2
        float ra = kDb2Lo[0] * q[0].x, rv = kDb2Hi[0] * q[0].x;       // band_a, band_v (adm_tools.c:336-361)
        float rh = kDb2Lo[0] * q[0].y, rd = kDb2Hi[0] * q[0].y;       // band_h, band_d (adm_tools.c:363-388)
        float da = kDb2Lo[0] * q[0].z, dv = kDb2Hi[0] * q[0].z;
        float dh = kDb2Lo[0] * q[0].w, dd = kDb2Hi[0] * q[0].w;
#pragma unroll
        for (int t = 1; t < 4; ++t) {
            ra = db2_acc<FAST>(ra, kDb2Lo[t], q[t].x);   rv = db2_acc<FAST>(rv, kDb2Hi[t], q[t].x);
            rh = db2_acc<FAST>(rh, kDb2Lo[t], q[t].y);   rd = db2_acc<FAST>(rd, kDb2Hi[t], q[t].y);
            da = db2_acc<FAST>(da, kDb2Lo[t], q[t].z);   dv = db2_acc<FAST>(dv, kDb2Hi[t], q[t].z);
            dh = db2_acc<FAST>(dh, kDb2Lo[t], q[t].w);   dd = db2_acc<FAST>(dd, kDb2Hi[t], q[t].w);
        }
        const bool interior = lane_on && tx >= 1 && tx <= ADM_TW && ty >= 1 && ty <= ADM_TH && cxr < bw && cyr < bh;
        if (interior && ll_ref) {
            const size_t g = (size_t)blockIdx.z * ll_frame_stride + (size_t)cyr * ll_row_stride + cxr;
            ll_ref[g] = ra;
            ll_dis[g] = da;
        }
        // adm_decouple (adm_tools.c:101-173), band order h, v, d
        const float o3[3] = { rh, rv, rd }, t3[3] = { dh, dv, dd };
        float rest[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float k = t3[b] / (o3[b] + 1e-30f);
            k = k < 0.0f ? 0.0f : (k > 1.0f ? 1.0f : k);
            rest[b] = k * o3[b];
        }
        const float dp = rh * dh + rv * dv;
        const float om = rh * rh + rv * rv;
        const float tm = dh * dh + dv * dv;
        const bool aligned = (dp >= 0.0f) && (dp * dp >= cos1sq * om * tm);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const float rf = (b == 2) ? rfac_d : rfac_hv;                 // adm_csf, adm_tools.c:190-202
            const float r = aligned ? t3[b] : rest[b];
            const float a = t3[b] - r;
            keep_o[s][b] = rf * o3[b];
            keep_r[s][b] = rf * r;
            const float aa = fabsf(rf * a);
            p30[s][b] = (1.0f / 30.0f) * aa;                              // adm_cm_thresh's weight of a neighbour (to LDS)
            p15[s][b] = (1.0f / 15.0f) * aa;                              // and of the centre (stays here)
        }
    }
    __syncthreads();                                                      // every float4 of vbuf has been read

    // (1/30)|csf_a| planes; slots on the image edge also fill the halo slots that mirror onto them
    // (adm_tools.c:241-248: -1 -> 1, n -> n-1), so that phase 3 indexes its 3x3 window blindly
    if (inner_tile) {
        if (lane_on) {
#pragma unroll
            for (int s = 0; s < ADM_SPT; ++s) {
#pragma unroll
                for (int b = 0; b < 3; ++b) absa[(b * ADM_SH + wv * ADM_SPT + s) * ADM_AW + tx] = p30[s][b];
            }
        }
    } else if (lane_on && cxr >= 0 && cxr < bw) {                        // slots outside the image own nothing
        const int cols[3] = { tx, (cxr == 1 && tx == 2) ? 0 : -1, (cxr == bw - 1 && tx + 1 < ADM_AW) ? tx + 1 : -1 };
#pragma unroll
        for (int s = 0; s < ADM_SPT; ++s) {
            const int ty = wv * ADM_SPT + s;
            const int cyr = cy0 - 1 + ty;
            const bool real = cyr >= 0 && cyr < bh;
            const int rows[3] = { real ? ty : -1, (cyr == 1 && ty == 2) ? 0 : -1, (cyr == bh - 1 && ty + 1 < ADM_SH) ? ty + 1 : -1 };
#pragma unroll
            for (int ri = 0; ri < 3; ++ri)
#pragma unroll
                for (int ci = 0; ci < 3; ++ci)
                    if (rows[ri] >= 0 && cols[ci] >= 0) {
#pragma unroll
                        for (int b = 0; b < 3; ++b) absa[(b * ADM_SH + rows[ri]) * ADM_AW + cols[ci]] = p30[s][b];
                    }
        }
    }
    __syncthreads();

    // ---- phase 3: masking threshold, contrast masking and cube sums for the interior slots
    {
        const int r0 = wv * ADM_SPT - 1;                                  // first neighbour row of this thread's slots
        const int xl = max(tx - 1, 0), xr = min(tx + 1, ADM_AW - 1);
        float thr[ADM_SPT];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const float *pl = absa + b * ADM_SH * ADM_AW;
            float nb[ADM_SPT + 2][3];
#pragma unroll
            for (int r = 0; r < ADM_SPT + 2; ++r) {
                const int rr = min(max(r0 + r, 0), ADM_SH - 1);           // clamped rows belong to halo slots only
                nb[r][0] = pl[rr * ADM_AW + xl];
                nb[r][1] = pl[rr * ADM_AW + tx];
                nb[r][2] = pl[rr * ADM_AW + xr];
            }
#pragma unroll
            for (int s = 0; s < ADM_SPT; ++s) {                           // adm_cm_thresh, adm_tools.c:220-257, same order of additions
                float a = nb[s][0];
                a = a + nb[s][1];
                a = a + nb[s][2];
                a = a + nb[s + 1][0];
                a = a + p15[s][b];
                a = a + nb[s + 1][2];
                a = a + nb[s + 2][0];
                a = a + nb[s + 2][1];
                a = a + nb[s + 2][2];
                thr[s] = b == 0 ? a : thr[s] + a;
            }
        }
#pragma unroll
        for (int s = 0; s < ADM_SPT; ++s) {
            const int ty = wv * ADM_SPT + s;
            const int cyr = cy0 - 1 + ty;
            const bool inwin = tx >= 1 && tx <= ADM_TW && ty >= 1 && ty <= ADM_TH &&
                               cxr >= left && cxr < right && cyr >= top && cyr < bottom;
            if (inwin) {
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    float x = fabsf(keep_r[s][b]) - thr[s];               // adm_cm, adm_tools.c:278-284
                    x = x < 0.0f ? 0.0f : x;
                    acc[b] += (double)(x * x * x);                        // adm_sum_cube, adm_tools.c:68-70
                    const float y = fabsf(keep_o[s][b]);
                    acc[3 + b] += (double)(y * y * y);
                }
            }
        }
    }
    }   // in_window
    }   // in_window || ll_ref
    const size_t tile = (size_t)blockIdx.z * gridDim.x + tile_id;
    block_reduce_store_lds<6, ADM_THREADS>(acc, red, partial + tile * 6);    // red aliases vbuf: 12 KB of its 36 KB
}

// ---------------------------------------------------------------- motion
// combo.c:309-326 + motion.c:44-60: 5-tap blur of the offset ref (convolution.c:78-93, y then x, in the selected tap
// order), SAD against the previous frame's blurred picture - summed in the reference's own order: one fp32 accumulator
// per row, columns left to right (motion.c:48-55), then the row sums top to bottom in fp32 (motion.c:56), so the
// `motion` atom is bit-identical to the reference's.  motion is the one atom whose plane sum reaches the model directly
// (VIF and ADM only feed ratios), and on periodic content a differently ordered sum is visibly different.
//
// A row's sum is a chain of w dependent additions; the rows and frames are independent chains.  So a WAVE owns a band of
// 8 rows x a chunk of consecutive frames and marches over the picture from left to right in steps of 60 columns; at
// every step it walks through its frames (the previous frame's blurred 8x60 tile stays in registers, as the blurred
// planes themselves never go to HBM).  No workgroup barrier anywhere: the four waves of a workgroup (32 rows) only share
// their halo rows through the L1.
//   vertical pass    lane = one of the 64 staged columns (60 + 2x2 halo) x 8 rows; its 12 input rows of the NEXT step
//                    are in flight while the current one is being filtered
//   horizontal pass  lane = (row, 8 consecutive outputs), read from the wave's LDS rows as three float4
//   row chains       the |differences| of four frames wait in the wave's LDS; then lane = (frame, row) adds the 60 values
//                    of its row one after the other to the row's running sum, which waits in LDS for the next 60
//                    columns (32 lanes at work; handing the sum from lane to lane inside a row with DPP moves instead
//                    keeps 8 lanes at work and costs 72 instructions per frame instead of 38)
// The chunk's first iteration per step only rebuilds the blur of the frame before the chunk (chunk 0 takes it from the
// plane carried over from the previous call); the last frame's tiles go to carry_out.  motion_total_kernel then adds
// the row sums of every frame in order.
constexpr int MO_TW = 60, MO_THREADS = 256;
constexpr int MO_CW = MO_TW + 4;                             // staged columns x0-2 .. x0+61
constexpr int MO_WAVES = MO_THREADS / 64;
constexpr int MO_VP = MO_CW + 4;                             // LDS row pitch: successive rows start 4 banks apart
constexpr int MO_CMAX = 8;                                   // longest chunk of frames (running sums in LDS)
constexpr int MO_AP = MO_TW + 1;                             // row pitch of the parked |differences|: odd, so that the 32 lanes of
                                                             // a chain step read 32 different banks
static_assert(MO_CW == 64, "one lane per staged column");

// Per-wave state of one step (60 columns) of the march
template <int OPT> struct MotionStep {
    int x0;
    bool vcol_avx, v_all_avx, h_all_avx;
    float prev[OPT], live[OPT];
};

// BH = rows per wave: 8 (lane = row x 8 strips of 8 outputs; chains of 4 frames) or, for launches that would otherwise have
// too few waves (4K, short batches: the launch's parallelism is rows x frames / chunk, nothing else), 4 (lane = row x 16
// strips of 4 outputs, the last strip idle; chains of 8 frames) - twice the waves, half the work per wave and iteration.
template <int ORDER, typename T, int BH>
__global__ __launch_bounds__(MO_THREADS) void motion_kernel(const T *__restrict__ ref, size_t frame_stride_bytes, int row_stride_bytes,
                                                            int w, int h, int nframes, int chunk, int has_carry,
                                                            const float *__restrict__ carry_in, float *__restrict__ carry_out,
                                                            float *__restrict__ rowsum, size_t rowsum_stride)
{
    constexpr int MO_BH = BH, MO_OPT = BH, MO_G = 32 / BH, MO_TH = BH * MO_WAVES;
    constexpr int SPR = 64 / BH;                                      // strips (lanes) per row in the horizontal pass
    static_assert(BH == 8 || BH == 4, "rows per wave");
    __shared__ __attribute__((aligned(16))) float v_all[MO_WAVES][MO_BH * MO_VP];
    __shared__ float acc_all[MO_WAVES][MO_CMAX * MO_BH];
    __shared__ float ad_all[MO_WAVES][MO_G * MO_BH * MO_AP];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *vb = v_all[wv], *acc_s = acc_all[wv], *adb = ad_all[wv];
    // (band, chunk) from the flat workgroup index, bands fastest, through the XCD-aware order: the workgroups one XCD receives
    // are vertical neighbours of one chunk, whose halo rows then meet in that XCD's L2
    const int item = xcd_tile_index(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);
    const int ck = __builtin_amdgcn_readfirstlane(item / (int)gridDim.x), band = item - ck * (int)gridDim.x;
    const int vy0 = band * MO_TH + wv * MO_BH;                        // first row of this wave (wave-uniform)
    if (vy0 >= h) return;                                             // no barrier below: a wave may leave alone
    const int f_begin = ck * chunk, f_end = min(nframes, f_begin + chunk);
    // warm-up iteration: blur only.  (readfirstlane: the compiler forms a saturating subtraction, which only the vector
    // ALU has, and every buffer load of a frame addressed through it would go through a waterfall loop)
    const int f_first = __builtin_amdgcn_readfirstlane(f_begin == 0 ? 0 : f_begin - 1);

    int roff[MO_BH + 4];                                              // byte offsets of the 12 input rows, wave-uniform (SGPRs)
#pragma unroll
    for (int k = 0; k < MO_BH + 4; ++k) roff[k] = mirror_idx(vy0 - 2 + k, h) * row_stride_bytes;
    const unsigned frame_bytes = (unsigned)h * (unsigned)row_stride_bytes;
    const unsigned char *ref_b = reinterpret_cast<const unsigned char *>(ref);
    // horizontal-pass identity
    const int hr = lane / SPR, strip = lane & (SPR - 1), ox0 = strip * MO_OPT;
    const int gy = vy0 + hr;
    const int hend = avx_hcol_end(w, 2);

    for (int i = lane; i < MO_CMAX * MO_BH; i += 64) acc_s[i] = 0.0f;   // motion.c:49: accum_line = 0

    // The wave's work is a sequence of iterations (step, frame): steps left to right, inside a step the frames f_first ..
    // f_end - 1.  The input rows of an iteration are requested TWO iterations ahead into one of two register sets (the
    // launch has few waves per SIMD - its parallelism is rows x chunks, the width is serial - so a wave has to keep
    // the memory system busy by itself: with one iteration of look-ahead an iteration took as long as a load)
    int ld_x0 = 0, ld_f = f_first;                                    // load cursor (wave-uniform)
    auto request = [&](float (&in)[MO_BH + 4]) {
        // unconditional (past the last iteration the cursor stays on the last step and the rows are loaded for nothing): with
        // loads under a branch the compiler cannot count the requests in flight and waits for all of them at the first use
        const int xc = mirror_idx(min(ld_x0, w - 1) - 2 + lane, w);
        const auto b = sample_rsrc(reinterpret_cast<const T *>(ref_b + (size_t)ld_f * frame_stride_bytes), frame_bytes);
#pragma unroll
        for (int k = 0; k < MO_BH + 4; ++k) in[k] = load_px_offset<T>(b, xc, roff[k]);           // offset_image, combo.c:236
        if (++ld_f == f_end) { ld_f = f_first; ld_x0 += MO_TW; }
    };

    MotionStep<MO_OPT> st;
    int f = f_first;
    st.x0 = 0;
    auto begin_step = [&]() {
        // the usual step has every sample in the vector order (convolution_avx.c:799-836 leaves it in the first and last two
        // rows, the columns from floor8(w) on in the vertical pass, the first two and the last few in the horizontal pass):
        // decided once per step and wave, so that the frames' loop carries no per-sample masks
        const int xc = mirror_idx(st.x0 - 2 + lane, w);
        st.vcol_avx = ORDER == ORDER_AVX && avx_vcol_ok(xc, w);
        bool hall = true;
#pragma unroll
        for (int j = 0; j < MO_OPT; ++j) {
            const int gx = st.x0 + ox0 + j;
            const bool mine = ox0 + j < MO_TW && gx < w && gy < h;
            st.live[j] = mine ? 1.0f : 0.0f;
            st.prev[j] = (f_begin == 0 && has_carry && mine) ? carry_in[(size_t)gy * w + gx] : 0.0f;
            hall = hall && gx >= 2 && gx < hend;
        }
        st.v_all_avx = __all(st.vcol_avx) && avx_row_ok(vy0, h, 2) && avx_row_ok(vy0 + MO_BH - 1, h, 2);
        st.h_all_avx = __all(hall);
    };
    // one iteration; returns false after the last one
    auto iterate = [&](float (&in)[MO_BH + 4]) -> bool {
        if (f == f_first) begin_step();
        wave_lds_sync();                                                              // the previous iteration's reads stay before these writes
        if (ORDER != ORDER_AVX || st.v_all_avx) {
#pragma unroll
            for (int i = 0; i < MO_BH; ++i) vb[i * MO_VP + lane] = taps_sel<5, true, ORDER>(in + i, 1, true);
        } else {
#pragma unroll
            for (int i = 0; i < MO_BH; ++i)
                vb[i * MO_VP + lane] = taps_sel<5, true, ORDER>(in + i, 1, st.vcol_avx && avx_row_ok(vy0 + i, h, 2));
        }
        wave_lds_sync();
        request(in);                                                                  // this set is free: the iteration after the next
        float win[MO_OPT + 4];
        {
            const float4 *src = reinterpret_cast<const float4 *>(vb + hr * MO_VP + ox0);
            // the last strip of a row reaches beyond the 64 staged columns: its outputs there are not the tile's (masked)
#pragma unroll
            for (int m = 0; m < (MO_OPT + 4) / 4; ++m) {
                const float4 q = (ox0 + 4 * m < MO_CW) ? src[m] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                win[4 * m] = q.x; win[4 * m + 1] = q.y; win[4 * m + 2] = q.z; win[4 * m + 3] = q.w;
            }
        }
        const bool scored = f >= f_begin && (f > 0 || has_carry);
        float cur[MO_OPT], ad[MO_OPT];
        if (ORDER != ORDER_AVX || st.h_all_avx) {
#pragma unroll
            for (int j = 0; j < MO_OPT; ++j) cur[j] = taps_sel<5, true, ORDER>(win + j, 1, true);
        } else {
#pragma unroll
            for (int j = 0; j < MO_OPT; ++j) cur[j] = taps_sel<5, true, ORDER>(win + j, 1, st.x0 + ox0 + j >= 2 && st.x0 + ox0 + j < hend);
        }
#pragma unroll
        for (int j = 0; j < MO_OPT; ++j) {
            // columns beyond the tile or the picture add +0.0f (x * 0.0f: the blurred values are finite), which leaves a
            // non-negative running sum as it is; x * 1.0f is x
            ad[j] = fabsf(st.prev[j] - cur[j]) * (scored ? st.live[j] : 0.0f);
            st.prev[j] = cur[j];
        }
        if (f >= f_begin) {
            // park the |differences| of this frame; every MO_G frames the wave turns to the row chains
            const int fi = f - f_begin, slot = fi & (MO_G - 1);
            float *dst = adb + (slot * MO_BH + hr) * MO_AP + ox0;
#pragma unroll
            for (int j = 0; j < MO_OPT; ++j)
                if (ox0 + j < MO_TW) dst[j] = ad[j];
            if (slot == MO_G - 1 || f + 1 == f_end) {
                // motion.c:50-55: accum_line += fabs(..) over the columns in order.  (The reference adds in double and
                // rounds to float at every step; for one addition of two floats that is the fp32 sum: 53 >= 2*24 + 2.)
                // lane = (frame of the group, row): its running sum from the previous 60 columns, then 60 additions
                wave_lds_sync();
                if (lane < (slot + 1) * MO_BH) {
                    const int ai = (fi - slot) * MO_BH + lane;
                    const float *src = adb + lane * MO_AP;
                    float t = acc_s[ai];
#pragma unroll
                    for (int j = 0; j < MO_TW; ++j) t = t + src[j];
                    acc_s[ai] = t;
                }
                wave_lds_sync();
            }
        }
        if (++f == f_end) {
            if (carry_out && f_end == nframes) {
#pragma unroll
                for (int j = 0; j < MO_OPT; ++j) {
                    const int gx = st.x0 + ox0 + j;
                    if (ox0 + j < MO_TW && gx < w && gy < h) carry_out[(size_t)gy * w + gx] = st.prev[j];
                }
            }
            f = f_first;
            st.x0 += MO_TW;
        }
        return st.x0 < w;
    };

    float in_a[MO_BH + 4], in_b[MO_BH + 4];
    request(in_a);
    request(in_b);
    while (iterate(in_a) && iterate(in_b)) {}

    wave_lds_sync();
    for (int i = lane; i < (f_end - f_begin) * MO_BH; i += 64) {
        const int fi = i / MO_BH, r = i % MO_BH;
        if (vy0 + r < h) rowsum[(size_t)(f_begin + fi) * rowsum_stride + vy0 + r] = acc_s[i];
    }
}

// motion.c:56-59: accum += accum_line over the rows in order, fp32.  One wave per frame: 64 row sums per load, added one
// after the other (every lane keeps the same running sum).  The total goes to the frame's first partial slot as a double.
__global__ __launch_bounds__(64) void motion_total_kernel(const float *__restrict__ rowsum, size_t rowsum_stride, int h,
                                                          double *__restrict__ total, size_t total_stride)
{
    const int lane = threadIdx.x;
    const float *rs = rowsum + (size_t)blockIdx.x * rowsum_stride;
    float acc = 0.0f;
    for (int r0 = 0; r0 < h; r0 += 64) {
        const float v = r0 + lane < h ? rs[r0 + lane] : 0.0f;        // + 0.0f beyond the last row: no change
#pragma unroll
        for (int i = 0; i < 64; ++i) acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i));
    }
    if (lane == 0) total[(size_t)blockIdx.x * total_stride] = (double)acc;
}

// ---------------------------------------------------------------- raw luma ingest
// file_io.c:86-172 on the device: 8-bit samples -> (float)px, 16-bit little-endian words -> (float)px / 4.0
// (exact: a power of two; no masking to 10 bits, as in the reference), dense fp32 planes out.  The -128 offset
// stays where it is for fp32 input: in the loads of the feature kernels.  Four samples per thread.
template <typename T>
__global__ __launch_bounds__(256) void widen_kernel(const unsigned char *__restrict__ src, size_t src_row_bytes, size_t src_frame_bytes,
                                                    float *__restrict__ dst, int w, int h, int vec_ok)
{
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x4 >= w) return;
    const T *row = reinterpret_cast<const T *>(src + (size_t)blockIdx.z * src_frame_bytes + (size_t)y * src_row_bytes);
    float *out = dst + ((size_t)blockIdx.z * h + y) * w;
    const float scale = sizeof(T) == 2 ? 0.25f : 1.0f;
    if (vec_ok && x4 + 4 <= w) {
        float4 v;
        if (sizeof(T) == 1) {
            const uchar4 p = *reinterpret_cast<const uchar4 *>(row + x4);
            v = make_float4((float)p.x, (float)p.y, (float)p.z, (float)p.w);
        } else {
            const ushort4 p = *reinterpret_cast<const ushort4 *>(row + x4);
            v = make_float4((float)p.x * scale, (float)p.y * scale, (float)p.z * scale, (float)p.w * scale);
        }
        *reinterpret_cast<float4 *>(out + x4) = v;
    } else {
        for (int x = x4; x < min(x4 + 4, w); ++x) out[x] = (float)row[x] * scale;
    }
}

// ---------------------------------------------------------------- optional metrics: PSNR, SSIM, MS-SSIM
// combo.c:168-230 -> psnr.c, ssim.c, ms_ssim.c, iqa/*.  They work on the UN-OFFSET planes.  The reference keeps
// float products and double running sums (iqa/convolve.c with IQA_CONVOLVE_1D: horizontal pass, float store,
// vertical pass, float store); the kernels keep the same types expression by expression.
constexpr int PSNR_ROWS = 8;

// psnr.c:37-50: sum of (float)(diff * diff) in double; one workgroup per PSNR_ROWS rows
__global__ __launch_bounds__(256) void psnr_kernel(const float *__restrict__ ref, const float *__restrict__ dis, size_t frame_stride,
                                                   int row_stride, int w, int h, double *__restrict__ partial)
{
    __shared__ double red[4];
    const float *fr = ref + (size_t)blockIdx.z * frame_stride, *fd = dis + (size_t)blockIdx.z * frame_stride;
    double acc[1] = { 0.0 };
    const int y1 = min(h, (int)(blockIdx.x + 1) * PSNR_ROWS);
    for (int y = blockIdx.x * PSNR_ROWS; y < y1; ++y)
        for (int x = threadIdx.x; x < w; x += 256) {
            const float diff = fr[(size_t)y * row_stride + x] - fd[(size_t)y * row_stride + x];
            acc[0] += (double)(diff * diff);
        }
    block_reduce_store<1, 4>(acc, red, partial + (size_t)blockIdx.z * gridDim.x + blockIdx.x);
}

// KBND_SYMMETRIC, iqa/convolve.c:35-42 (clamped for workgroups hanging over the plane)
__device__ __forceinline__ int sym_idx(int i, int n)
{
    i = i < 0 ? -1 - i : i;
    i = i >= n ? 2 * n - i - 1 : i;
    return min(max(i, 0), n - 1);
}

// ssim.c:117-163: _iqa_decimate by `f` with the f x f box kernel of 1.0f/(f*f) (2-D form, iqa/convolve.c:219-268,
// iqa/decimate.c:37-59): out (x, y) = kernel centred (even f: one tap more to the left/top) on (x*f, y*f).
__global__ __launch_bounds__(256) void box_decimate_kernel(const float *__restrict__ src, size_t src_frame_stride, int src_row_stride,
                                                           int w, int h, int f, float *__restrict__ dst, int sw, int sh)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= sw || y >= sh) return;
    const float *img = src + (size_t)blockIdx.z * src_frame_stride;
    const int uc = f / 2, even = (f & 1) ? 0 : 1;
    const float kv = 1.0f / (f * f);
    double sum = 0.0;
    for (int v = -uc; v <= uc - even; ++v) {
        const float *row = img + (size_t)sym_idx(y * f + v, h) * src_row_stride;
        for (int u = -uc; u <= uc - even; ++u) sum += (double)(row[sym_idx(x * f + u, w)] * kv);
    }
    dst[((size_t)blockIdx.z * sh + y) * sw + x] = (float)sum;
}

// The same decimation for f <= BOXR_FMAX with coalesced reads: a wave forms 64 neighbouring outputs of one output row.  Per
// kernel row it loads the 64 * f consecutive input samples those outputs cover (lane l takes samples l, l + 64, ...), parks
// them in its quarter of the LDS (one float of padding after every 32: the strided reads that follow then hit 32 different
// banks) and every lane reads its own f samples back.  box_decimate_kernel lets each lane fetch its window from memory
// directly, f dwords at a stride of 4 * f bytes per instruction (3 TB/s at 1080p; this one is bound by the plane read).
// blockIdx.z = 2 * frame + plane: both planes of a pair in one launch.  Same sum: row-major, float products, double sum.
constexpr int BOXR_FMAX = 8;
constexpr int BOXR_ROW = 64 * BOXR_FMAX + 64 * BOXR_FMAX / 32;             // floats of LDS per wave
__global__ __launch_bounds__(256) void box_decimate_rows_kernel(const float *__restrict__ ref, const float *__restrict__ dis,
                                                                size_t src_frame_stride, int src_row_stride, int w, int h, int f,
                                                                float *__restrict__ oref, float *__restrict__ odis, int sw, int sh)
{
    __shared__ float stage[4][BOXR_ROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frame = blockIdx.z >> 1, plane = blockIdx.z & 1;
    const float *img = (plane ? dis : ref) + (size_t)frame * src_frame_stride;
    const int x = blockIdx.x * 64 + lane, y = blockIdx.y * 4 + wave;
    const int uc = f / 2;
    const float kv = 1.0f / (f * f);
    const int i0 = blockIdx.x * 64 * f - uc;                                // first input column of the wave's span
    int col[BOXR_FMAX];
#pragma unroll
    for (int k = 0; k < BOXR_FMAX; ++k) col[k] = sym_idx(i0 + lane + 64 * k, w);
    float *mine = stage[wave];
    double sum = 0.0;
    for (int v = 0; v < f; ++v) {                                           // the same trip count in every wave: barriers inside
        const float *row = img + (size_t)sym_idx(y * f - uc + v, h) * src_row_stride;
        float a[BOXR_FMAX];
#pragma unroll
        for (int k = 0; k < BOXR_FMAX; ++k)
            if (k < f) a[k] = row[col[k]];
        __syncthreads();                                                    // the previous row's reads are done
#pragma unroll
        for (int k = 0; k < BOXR_FMAX; ++k)
            if (k < f) {
                const int j = lane + 64 * k;
                mine[j + (j >> 5)] = a[k];
            }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < BOXR_FMAX; ++u)
            if (u < f) {
                const int j = lane * f + u;
                sum += (double)(mine[j + (j >> 5)] * kv);
            }
    }
    if (x < sw && y < sh) (plane ? odis : oref)[((size_t)frame * sh + y) * sw + x] = (float)sum;
}

// ms_ssim.c:41-51: 9/7 biorthogonal low-pass, the 2-D table _iqa_decimate applies (81 float taps, row-major),
// widened to double and kept in memory: the kernel row loop is not unrolled and fetches its 9 coefficients through the
// scalar cache (81 double literals would not fit the SGPR file and end up in VGPRs)
__device__ const double kLpf9D[81] = {
#define L9(x) (double)x##f
     L9(0.000714),L9(-0.000450),L9(-0.002090), L9(0.007132), L9(0.016114), L9(0.007132),L9(-0.002090),L9(-0.000450), L9(0.000714),
    L9(-0.000450), L9(0.000283), L9(0.001316),L9(-0.004490),L9(-0.010146),L9(-0.004490), L9(0.001316), L9(0.000283),L9(-0.000450),
    L9(-0.002090), L9(0.001316), L9(0.006115),L9(-0.020867),L9(-0.047149),L9(-0.020867), L9(0.006115), L9(0.001316),L9(-0.002090),
     L9(0.007132),L9(-0.004490),L9(-0.020867), L9(0.071207), L9(0.160885), L9(0.071207),L9(-0.020867),L9(-0.004490), L9(0.007132),
     L9(0.016114),L9(-0.010146),L9(-0.047149), L9(0.160885), L9(0.363505), L9(0.160885),L9(-0.047149),L9(-0.010146), L9(0.016114),
     L9(0.007132),L9(-0.004490),L9(-0.020867), L9(0.071207), L9(0.160885), L9(0.071207),L9(-0.020867),L9(-0.004490), L9(0.007132),
    L9(-0.002090), L9(0.001316), L9(0.006115),L9(-0.020867),L9(-0.047149),L9(-0.020867), L9(0.006115), L9(0.001316),L9(-0.002090),
    L9(-0.000450), L9(0.000283), L9(0.001316),L9(-0.004490),L9(-0.010146),L9(-0.004490), L9(0.001316), L9(0.000283),L9(-0.000450),
     L9(0.000714),L9(-0.000450),L9(-0.002090), L9(0.007132), L9(0.016114), L9(0.007132),L9(-0.002090),L9(-0.000450), L9(0.000714),
#undef L9
};
constexpr int LPF_TW = 64, LPF_TH = 16, LPF_OPT = 4;                       // outputs per workgroup, outputs per thread (one row)
constexpr int LPF_IW = 2 * LPF_TW + 8, LPF_IH = 2 * LPF_TH + 7;            // staged 136 x 39 samples
constexpr int LPF_NW = 2 * (LPF_OPT - 1) + 9;                               // a thread's window per kernel row: 15 samples
constexpr int LPF_GP = 12;                                                  // LDS: every group of 8 samples is followed by 4 floats of padding:
constexpr int LPF_RP = (LPF_IW / 8) * LPF_GP;                               // neighbouring threads' windows then start 48 bytes apart and the
                                                                            // float4 reads of 8 lanes cover all 32 banks
static_assert(LPF_TH * (LPF_TW / LPF_OPT) == 256 && LPF_IW % 8 == 0 && LPF_NW <= 16, "thread = (output row, strip of 4 outputs)");

// ms_ssim.c:250-266: one level of the pyramid, both planes of a frame pair handled by blockIdx.z = 2*frame + plane.
// _iqa_decimate -> _iqa_filter_pixel (iqa/convolve.c:219-268): sum over the 9x9 table in row-major order of img * k, the
// product rounded to float, the sum kept in double, the result stored as float.  As in ssim_taps the terms enter the double
// sum through v_fma_f64 (exact products; see there); a thread forms four neighbouring outputs of a row from one 16-sample
// window per kernel row (widened once) instead of 9 samples per output.
// EXACT: the reference's own expression - every product rounded to float before it is widened and added (three
// instructions per tap instead of one); vmaf_hip_engine_set_metrics_exact.
template <bool EXACT>
__global__ __launch_bounds__(256, 4) void lpf_decimate_kernel(const float *__restrict__ ref, const float *__restrict__ dis,
                                                           size_t src_frame_stride, int src_row_stride, int w, int h,
                                                           float *__restrict__ oref, float *__restrict__ odis, int sw, int sh)
{
    __shared__ __attribute__((aligned(16))) float tile[LPF_IH * LPF_RP];
    const int frame = blockIdx.z >> 1, plane = blockIdx.z & 1;
    const float *img = (plane ? dis : ref) + (size_t)frame * src_frame_stride;
    float *out = (plane ? odis : oref) + (size_t)frame * sw * sh;
    const int x0 = blockIdx.x * LPF_TW, y0 = blockIdx.y * LPF_TH;
    {   // staging: thread = one of the first 128 tile columns x every second tile row; the row offset is wave-uniform
        // (an SGPR), the mirrored column index is computed once (buffer loads, as in the feature kernels).  All loads of a
        // thread are in flight before its first LDS write.
        const int tid = threadIdx.x;
        const int half = __builtin_amdgcn_readfirstlane(tid >> 7), lx = tid & 127;
        const auto br = frame_rsrc(img, (unsigned)h * (unsigned)src_row_stride * 4u);
        const int gx = sym_idx(2 * x0 - 4 + lx, w);
        constexpr int NR = (LPF_IH + 1) / 2;
        float a[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int ly = 2 * i + half;
            a[i] = buf_load(br, gx * 4, sym_idx(2 * y0 - 4 + ly, h) * src_row_stride * 4);   // ly == LPF_IH (half = 1, last i): a valid row, not stored
        }
        // the last 8 columns of the tile: 39 x 8 samples, one or two per thread
        float e[2];
        int epos[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + 256 * j;
            epos[j] = -1;
            e[j] = 0.0f;
            if (idx < LPF_IH * 8) {
                const int ly = idx >> 3, ex = 128 + (idx & 7);
                e[j] = img[(size_t)sym_idx(2 * y0 - 4 + ly, h) * src_row_stride + sym_idx(2 * x0 - 4 + ex, w)];
                epos[j] = ly * LPF_RP + (ex >> 3) * LPF_GP + (ex & 7);
            }
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int ly = 2 * i + half;
            if (ly < LPF_IH) tile[ly * LPF_RP + (lx >> 3) * LPF_GP + (lx & 7)] = a[i];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (epos[j] >= 0) tile[epos[j]] = e[j];
    }
    __syncthreads();
    const int oy = threadIdx.x >> 4, seg = threadIdx.x & 15, ox0 = seg * LPF_OPT;
    double sum[LPF_OPT] = {};
#pragma unroll 1
    for (int v = 0; v < 9; ++v) {
        double win[16];
        // samples 8*seg .. 8*seg + 15 of the tile row = groups seg and seg + 1
        const float4 *src = reinterpret_cast<const float4 *>(tile + (2 * oy + v) * LPF_RP + seg * LPF_GP);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float4 f = src[(m >> 1) * (LPF_GP / 4) + (m & 1)];
            win[4 * m] = (double)f.x; win[4 * m + 1] = (double)f.y; win[4 * m + 2] = (double)f.z; win[4 * m + 3] = (double)f.w;
        }
#pragma unroll
        for (int u = 0; u < 9; ++u)
#pragma unroll
            for (int o = 0; o < LPF_OPT; ++o) {
                if (EXACT) sum[o] = sum[o] + (double)((float)win[2 * o + u] * (float)kLpf9D[v * 9 + u]);   // iqa/convolve.c:262
                else sum[o] = __builtin_fma(win[2 * o + u], kLpf9D[v * 9 + u], sum[o]);
            }
    }
#pragma unroll
    for (int o = 0; o < LPF_OPT; ++o)
        if (x0 + ox0 + o < sw && y0 + oy < sh) out[(size_t)(y0 + oy) * sw + x0 + ox0 + o] = (float)sum[o];
}

// ssim_tools.h: g_gaussian_window_h == g_gaussian_window_v
// Tile 32 x 48 outputs: the horizontal pass runs over the tile's 58 rows (58 / 48 = 1.21 x the outputs; a 64 x 16 tile paid
// 26 / 16 = 1.63 x) on 58 * 4 = 232 of the 256 threads.
constexpr int SS_TW = 32, SS_TH = 48, SS_IW = SS_TW + 10, SS_IH = SS_TH + 10, SS_IP = SS_IW + 2;   // staged 58 x 42 (pitch 44)
constexpr int SS_SEG = 8;                                      // outputs per thread in the horizontal pass (one row)
constexpr int SS_VR = 6;                                       // outputs per thread in the vertical pass (one column)
static_assert(SS_IH * (SS_TW / SS_SEG) <= 256 && SS_TW * (SS_TH / SS_VR) == 256, "thread maps of ssim_stat_kernel");

// 11-tap window sum.  The reference (iqa/convolve.c:113-143) rounds every product img * k to float and adds the products
// to a double running sum in tap order; the result is stored as float.  Here the running sum takes the products through
// v_fma_f64: the product of two floats is exact in double (48 bits), so each term enters the sum WITHOUT the reference's
// intermediate float rounding - one fp64 instruction per tap instead of v_mul_f32 + v_cvt_f64_f32 + v_add_f64 (fp64 issues
// at half the fp32 rate on gfx950: 4 cycles instead of 10 per tap), and the samples are widened once per window, not once
// per tap.  Same tap order, same double accumulator, same float result type; the window sums differ from the reference's
// in the last float bit in a minority of samples (they are the more exact ones), SSIM / MS-SSIM by < 1e-7
// (tests/test_gpu_parity.py::test_psnr_ssim_ms_ssim_match_oracle holds them to 1e-6 against the oracle, which keeps the
// reference's arithmetic expression by expression).
__device__ constexpr double kSsimWinD[11] = { (double)0.001028f, (double)0.007599f, (double)0.036001f, (double)0.109361f, (double)0.213006f,
                                              (double)0.266012f, (double)0.213006f, (double)0.109361f, (double)0.036001f, (double)0.007599f,
                                              (double)0.001028f };
// EXACT = the reference's expression itself: sum += img * k with the float product (iqa/convolve.c:129, :144); the window
// holds samples that are floats widened to double, so narrowing them again is exact.
template <int NOUT, bool EXACT> __device__ __forceinline__ void ssim_taps(const double *win, float *out)
{
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            if (EXACT) sum = sum + (double)((float)win[o + k] * (float)kSsimWinD[k]);
            else sum = __builtin_fma(win[o + k], kSsimWinD[k], sum);
        }
        out[o] = (float)sum;
    }
}

// 1 / d for d well inside the normal range (here: 29 .. 1e5): v_rcp_f64 and two Newton steps, relative error ~1e-16.  The
// compiler's IEEE division spends 12 fp64 instructions per quotient (scaling, fix-ups for denormals and infinities).
__device__ __forceinline__ double rcp_newton(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(e, r, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(e, r, r);
}

// sqrt((double)x) for a float x >= 0: v_sqrt_f32 (1 ulp) and one Newton step in double, whose residual x - s0 * s0 is exact
// (a 48-bit product): relative error ~1e-14; x = 0 gives 0.
__device__ __forceinline__ double sqrt_of_float(float x)
{
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float r0 = s0 > 0.0f ? __builtin_amdgcn_rcpf(s0) : 0.0f;
    const double sd = (double)s0;
    const double e = __builtin_fma(-sd, sd, (double)x);
    return __builtin_fma(e, 0.5 * (double)r0, sd);
}

// _iqa_ssim with default arguments (iqa/ssim_tools.c:92-243) on dense planes [frame][h][w]: "valid" 11-tap
// separable Gaussian of {ref, cmp, ref^2, cmp^2, ref*cmp} (horizontal pass first, iqa/convolve.c:113-143), sigma,
// l / c / s per pixel in double, four sums per workgroup: sum(l*c*s), sum(l), sum(c), sum(s).
// Horizontal pass: thread = (row, 8 outputs), the 18-sample windows of ref and cmp in registers, products formed
// there; vertical pass: thread = (column, 6 outputs) with a 16-sample register window per quantity.
// CS_ONLY: the MS-SSIM levels 0..3, whose luminance term carries the exponent 0 (ms_ssim.c:62-64: pow(l, 0) = 1 whatever l
// is): only sum(c) and sum(s) are formed, the l division is skipped and the l*c*s / l sums stay 0.
// PSNR: the launch on the full-resolution planes (MS-SSIM level 0) also forms the PSNR noise sum of psnr.c:37-50,
// sum (double)(float)(diff * diff) over EVERY pixel of the plane, from the samples it stages anyway (psnr_kernel reads both
// planes again: 0.36 ms per 120 frames of 1080p).  A workgroup adds the pixels of its own 32 x 48 rectangle; the tiles
// cover the valid area (w - 10) x (h - 10) only, so the last tile of a row / column also takes the up to 10 columns / rows
// behind its rectangle, which its staged halo holds.
// EXACT: the window sums and the per-pixel sqrt / divisions as the reference's expressions (IEEE double sqrt and division
// instead of the hardware estimates + Newton steps) - per-pixel l, c, s bit-identical to iqa/ssim_tools.c.
template <bool CS_ONLY, bool PSNR, bool EXACT>
__global__ __launch_bounds__(256, 3) void ssim_stat_kernel(const float *__restrict__ ref, const float *__restrict__ cmp, int w, int h,
                                                        double *__restrict__ partial, double *__restrict__ psnr_partial)
{
    // one LDS area, used twice: the staged input tiles (rt, ct) until every thread of the horizontal pass holds its
    // windows in registers, then the five horizontally filtered planes (hc) - 37 KB; 150 VGPRs: three workgroups per CU
    __shared__ __attribute__((aligned(16))) float smem[5 * SS_IH * SS_TW];
    static_assert(2 * SS_IH * SS_IP <= 5 * SS_IH * SS_TW, "the staged tiles fit the area of the filtered planes");
    float *rt = smem, *ct = smem + SS_IH * SS_IP;
    float (*hc)[SS_IH * SS_TW] = reinterpret_cast<float (*)[SS_IH * SS_TW]>(smem);
    __shared__ double red[16], red_psnr[4];
    const int tid = threadIdx.x;
    const int vw = w - 10, vh = h - 10;                              // valid output size
    const int x0 = blockIdx.x * SS_TW, y0 = blockIdx.y * SS_TH;
    const float *fr = ref + (size_t)blockIdx.z * w * h, *fc = cmp + (size_t)blockIdx.z * w * h;
    const size_t tile = (size_t)blockIdx.z * gridDim.x * gridDim.y + blockIdx.y * gridDim.x + blockIdx.x;
    {   // all global loads of a thread in flight before its first LDS write
        constexpr int N = SS_IH * SS_IW, IT = (N + 255) / 256;
        float a[IT], b[IT];
        double noise[1] = { 0.0 };
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int idx = tid + it * 256;
            if (idx < N) {
                const int ly = idx / SS_IW, lx = idx - ly * SS_IW;
                const size_t g = (size_t)min(y0 + ly, h - 1) * w + min(x0 + lx, w - 1);   // clamped reads feed masked outputs only
                a[it] = fr[g];
                b[it] = fc[g];
            }
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int idx = tid + it * 256;
            if (idx < N) {
                const int ly = idx / SS_IW, lx = idx - ly * SS_IW;
                rt[ly * SS_IP + lx] = a[it];
                ct[ly * SS_IP + lx] = b[it];
                if (PSNR) {
                    const bool mine = (lx < SS_TW || blockIdx.x == gridDim.x - 1) && (ly < SS_TH || blockIdx.y == gridDim.y - 1) &&
                                      x0 + lx < w && y0 + ly < h;
                    const float diff = a[it] - b[it];
                    if (mine) noise[0] += (double)(diff * diff);
                }
            }
        }
        if (PSNR) block_reduce_store<1, 4>(noise, red_psnr, psnr_partial + tile);   // its barrier is the one the tiles need
    }
    if (!PSNR) __syncthreads();
    const bool hthread = tid < SS_IH * (SS_TW / SS_SEG);
    const int row = hthread ? tid / (SS_TW / SS_SEG) : 0, seg = hthread ? tid - row * (SS_TW / SS_SEG) : 0;
    float a[SS_SEG + 10], b[SS_SEG + 10];
    {
        const float2 *pa = reinterpret_cast<const float2 *>(rt + row * SS_IP + seg * SS_SEG);
        const float2 *pb = reinterpret_cast<const float2 *>(ct + row * SS_IP + seg * SS_SEG);
#pragma unroll
        for (int m = 0; m < (SS_SEG + 10) / 2; ++m) {
            const float2 u = pa[m], v = pb[m];
            a[2 * m] = u.x; a[2 * m + 1] = u.y; b[2 * m] = v.x; b[2 * m + 1] = v.y;
        }
    }
    __syncthreads();                                                  // rt / ct are dead: hc may overwrite them
    if (hthread) {
        float o[SS_SEG];
        double t[SS_SEG + 10];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            // the squares and the cross product are float products, as in ssim_tools.c:147-151
#pragma unroll
            for (int k = 0; k < SS_SEG + 10; ++k)
                t[k] = (double)(q == 0 ? a[k] : q == 1 ? b[k] : q == 2 ? a[k] * a[k] : q == 3 ? b[k] * b[k] : a[k] * b[k]);
            ssim_taps<SS_SEG, EXACT>(t, o);
            float4 *dst = reinterpret_cast<float4 *>(&hc[q][row * SS_TW + seg * SS_SEG]);
            dst[0] = make_float4(o[0], o[1], o[2], o[3]);
            dst[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
    __syncthreads();
    const float K1 = 0.01f, K2 = 0.03f;
    const int L = 255;
    const float C1 = (K1 * L) * (K1 * L), C2 = (K2 * L) * (K2 * L);
    double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
    {
        const int ox = tid & (SS_TW - 1), oy0 = (tid / SS_TW) * SS_VR;
        float m[5][SS_VR];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            double win[SS_VR + 10];
#pragma unroll
            for (int k = 0; k < SS_VR + 10; ++k) win[k] = (double)hc[q][(oy0 + k) * SS_TW + ox];
            ssim_taps<SS_VR, EXACT>(win, m[q]);
        }
#pragma unroll
        for (int j = 0; j < SS_VR; ++j) {
            if (x0 + ox < vw && y0 + oy0 + j < vh) {
                const float ref_mu = m[0][j], cmp_mu = m[1][j];
                float rss = m[2][j], css = m[3][j], sb = m[4][j];
                rss -= ref_mu * ref_mu;
                css -= cmp_mu * cmp_mu;
                rss = (float)((0.0 > rss) ? 0.0 : rss);
                css = (float)((0.0 > css) ? 0.0 : css);
                sb -= ref_mu * cmp_mu;
                // the reference's sqrt() and divisions are IEEE double; the ones here agree with them to ~1e-14 relative
                // (a per-pixel term of size <= 1 entering a mean: far below the 1e-7 of the window sums)
                const double srsc = EXACT ? __builtin_sqrt((double)(rss * css)) : sqrt_of_float(rss * css);
                const double c = EXACT ? (2.0 * srsc + C2) / (double)(rss + css + C2) : (2.0 * srsc + C2) * rcp_newton((double)(rss + css + C2));
                const double s_ = EXACT ? (sb + C2 / 2.0) / (srsc + C2 / 2.0) : (sb + C2 / 2.0) * rcp_newton(srsc + C2 / 2.0);
                acc[2] += c; acc[3] += s_;
                if (!CS_ONLY) {
                    const double ld = (double)(ref_mu * ref_mu + cmp_mu * cmp_mu + C1);
                    const double l = EXACT ? (2.0 * ref_mu * cmp_mu + C1) / ld : (2.0 * ref_mu * cmp_mu + C1) * rcp_newton(ld);
                    acc[0] += l * c * s_;
                    acc[1] += l;
                }
            }
        }
    }
    block_reduce_store<4, 4>(acc, red, partial + tile * 4);
}

// ANSNR (ansnr.c:44-104 with the 2-D filters of ansnr_tools.c:36-58, :159-199): ref through the 3x3 binomial, dis
// through the 5x5 kernel, float sums per kernel row then over rows, mirrored borders; sig = sum ref_filtr^2,
// noise = sum (ref_filtr - dis_filtd)^2.  Runs on the offset planes like every caller of compute_ansnr does.
constexpr float kAnsnrRef[9] = { (float)(1.0 / 16.0), (float)(2.0 / 16.0), (float)(1.0 / 16.0), (float)(2.0 / 16.0), (float)(4.0 / 16.0),
                                 (float)(2.0 / 16.0), (float)(1.0 / 16.0), (float)(2.0 / 16.0), (float)(1.0 / 16.0) };
constexpr float kAnsnrDis[25] = {
    (float)(2.0 / 571.0),  (float)(7.0 / 571.0),  (float)(12.0 / 571.0),  (float)(7.0 / 571.0),  (float)(2.0 / 571.0),
    (float)(7.0 / 571.0),  (float)(31.0 / 571.0), (float)(52.0 / 571.0),  (float)(31.0 / 571.0), (float)(7.0 / 571.0),
    (float)(12.0 / 571.0), (float)(52.0 / 571.0), (float)(127.0 / 571.0), (float)(52.0 / 571.0), (float)(12.0 / 571.0),
    (float)(7.0 / 571.0),  (float)(31.0 / 571.0), (float)(52.0 / 571.0),  (float)(31.0 / 571.0), (float)(7.0 / 571.0),
    (float)(2.0 / 571.0),  (float)(7.0 / 571.0),  (float)(12.0 / 571.0),  (float)(7.0 / 571.0),  (float)(2.0 / 571.0) };
constexpr int AN_TW = 64, AN_TH = 16, AN_IW = AN_TW + 4, AN_IH = AN_TH + 4;

__global__ __launch_bounds__(256) void ansnr_kernel(const float *__restrict__ ref, const float *__restrict__ dis, size_t frame_stride,
                                                    int row_stride, int w, int h, double *__restrict__ partial)
{
    __shared__ float tr[AN_IH * AN_IW], td[AN_IH * AN_IW];
    __shared__ double red[8];
    const float *fr = ref + (size_t)blockIdx.z * frame_stride, *fd = dis + (size_t)blockIdx.z * frame_stride;
    const int x0 = blockIdx.x * AN_TW, y0 = blockIdx.y * AN_TH;
    {   // all global loads of a thread in flight before its first LDS write
        constexpr int N = AN_IH * AN_IW, IT = (N + 255) / 256;
        float a[IT], b[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int idx = threadIdx.x + it * 256;
            if (idx < N) {
                const int ly = idx / AN_IW, lx = idx - ly * AN_IW;
                const size_t g = (size_t)mirror_idx(y0 - 2 + ly, h) * row_stride + mirror_idx(x0 - 2 + lx, w);
                a[it] = fr[g];
                b[it] = fd[g];
            }
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int idx = threadIdx.x + it * 256;
            if (idx < N) {
                tr[idx] = a[it] + -128.0f;                           // offset_image, ansnr.c:174-177 / all.c:128-129
                td[idx] = b[it] + -128.0f;
            }
        }
    }
    __syncthreads();
    double acc[2] = { 0.0, 0.0 };
#pragma unroll
    for (int it = 0; it < AN_TW * AN_TH / 256; ++it) {
        const int ox = threadIdx.x & 63, oy = (threadIdx.x >> 6) + it * 4;
        float rf = 0.0f, df = 0.0f;
#pragma unroll
        for (int fi = 0; fi < 3; ++fi) {
            float inner = kAnsnrRef[fi * 3] * tr[(oy + 1 + fi) * AN_IW + ox + 1];
#pragma unroll
            for (int fj = 1; fj < 3; ++fj) inner = inner + kAnsnrRef[fi * 3 + fj] * tr[(oy + 1 + fi) * AN_IW + ox + 1 + fj];
            rf = fi == 0 ? inner : rf + inner;
        }
#pragma unroll
        for (int fi = 0; fi < 5; ++fi) {
            float inner = kAnsnrDis[fi * 5] * td[(oy + fi) * AN_IW + ox];
#pragma unroll
            for (int fj = 1; fj < 5; ++fj) inner = inner + kAnsnrDis[fi * 5 + fj] * td[(oy + fi) * AN_IW + ox + fj];
            df = fi == 0 ? inner : df + inner;
        }
        if (x0 + ox < w && y0 + oy < h) {
            acc[0] += (double)(rf * rf);
            acc[1] += (double)((rf - df) * (rf - df));
        }
    }
    block_reduce_store<2, 4>(acc, red, partial + ((size_t)blockIdx.z * gridDim.x * gridDim.y + blockIdx.y * gridDim.x + blockIdx.x) * 2);
}

// ---------------------------------------------------------------- final reduction
__global__ __launch_bounds__(256) void reduce_kernel(const double *__restrict__ partial, vk_reduce_table tab,
                                                     double *__restrict__ sums /* [frame][VK_NSUMS] */)
{
    __shared__ double red[4];
    const vk_reduce_desc d = tab.d[blockIdx.x];
    const int f = blockIdx.y;
    const double *p = partial + d.offset + (size_t)f * d.ntiles * d.k + d.comp;
    double acc[1] = { 0.0 };
    for (int t = threadIdx.x; t < d.ntiles; t += 256) acc[0] += p[(size_t)t * d.k];
    block_reduce_store<1, 4>(acc, red, sums + (size_t)f * VK_NSUMS + blockIdx.x);
}

// ---------------------------------------------------------------- nu-SVR RBF (svm.cpp:179-247)
__global__ __launch_bounds__(256) void svm_rbf_kernel(const double *__restrict__ sv, const double *__restrict__ coef, int n_sv,
                                                      int dim, double gamma, double rho, const double *__restrict__ x,
                                                      double *__restrict__ out)
{
    __shared__ double red[4];
    const double *xf = x + (size_t)blockIdx.x * dim;
    double acc[1] = { 0.0 };
    for (int i = threadIdx.x; i < n_sv; i += 256) {
        double d2 = 0.0;
        for (int j = 0; j < dim; ++j) {
            const double d = xf[j] - sv[(size_t)i * dim + j];
            d2 += d * d;
        }
        acc[0] += coef[i] * exp(-gamma * d2);
    }
    __shared__ double total;
    block_reduce_store<1, 4>(acc, red, &total);
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = total - rho;
}

// tiles flattened into grid.x (XCD-aware order inside the kernel), frames in grid.z
inline dim3 grid_flat(int w, int h, int tw, int th, int nframes) { return dim3(((w + tw - 1) / tw) * ((h + th - 1) / th), 1, nframes); }

}  // namespace

// ---------------------------------------------------------------- launchers
extern "C" {

// Planes at least 240 wide whose width is a multiple of 8 (no scalar-order columns but the two edge strips; the last tile
// of a row may be narrower) take the workgroup-wide tiles of vif_stat_wg_kernel; every other size the wave-private 48x32
// tiles.
static bool vif_use_wg(int w, int h) { return w >= VWG_TW && w % 8 == 0 && h >= 17; }
int vk_vif_tiles(int w, int h)
{
    if (vif_use_wg(w, h)) return ((w + VWG_TW - 1) / VWG_TW) * ((h + VWG_TH - 1) / VWG_TH);
    return ((w + VIF_TW - 1) / VIF_TW) * ((h + VIF_TH - 1) / VIF_TH);
}
int vk_adm_tiles(int w, int h)
{
    const int bw = (w + 1) / 2, bh = (h + 1) / 2;
    return ((bw + ADM_TW - 1) / ADM_TW) * ((bh + ADM_TH - 1) / ADM_TH);
}
// doubles per frame in the partial buffer: [0] the frame's SAD total (fp32 value), then its h row sums as packed floats
int vk_motion_tiles(int w, int h) { (void)w; return 1 + (h + 1) / 2; }

// The runtime loads a library's code object onto a device when the first of its kernels is launched there (or looked up):
// ~10 ms for this library, on the launching thread.  An engine being created asks for it here, on a helper thread, while
// its streams are set up - so that the first batch does not pay for it.  All kernels live in this one code object.
hipError_t vk_preload(void)
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&reduce_kernel));
}

hipError_t vk_launch_vif_stat(int scale, int order, const void *ref, const void *dis, int sample_bytes, size_t frame_stride, int row_stride,
                              int w, int h, int nframes, double *partial, float *oref, float *odis, size_t out_frame_stride,
                              int out_row_stride, float *nd, hipStream_t s)
{
    static const int fw[4] = { 17, 9, 5, 3 };
    if (sample_bytes != 4 && scale != 0) return hipErrorInvalidValue;      // only the picture itself can be raw luma
    const dim3 g = grid_flat(w, h, VIF_TW, VIF_TH, nframes);
    const BorderGeom bg = border_geom(w, h, fw[scale] / 2, VIF_TW, VIF_TH);
    const int nb = bg.ntiles, ni = (bg.tx_r - 1) * (bg.ty_b - 1), nb_pad = ni > 0 ? (nb + 7) / 8 * 8 : nb;
    const dim3 gm(nb_pad + ni, 1, nframes);                   // border-frame tiles (padded) first, then the interior rectangle
#define VIF_ARGS (const T *)ref, (const T *)dis, frame_stride, row_stride, w, h, partial, oref, odis, out_frame_stride, out_row_stride, nd
    const bool wg = vif_use_wg(w, h);
    const dim3 gw(wg ? vk_vif_tiles(w, h) : 1, 1, nframes);
#define VIF_STAT(FW, OFF, TYPE)                                                                        \
    do {                                                                                               \
        using T = TYPE;                                                                                \
        if (wg) {                                                                                      \
            if (order == ORDER_SCALAR) vif_stat_wg_kernel<FW, OFF, ORDER_SCALAR, T><<<gw, VIF_THREADS, 0, s>>>(VIF_ARGS);   \
            else if (order == ORDER_FMA) vif_stat_wg_kernel<FW, OFF, ORDER_FMA, T><<<gw, VIF_THREADS, 0, s>>>(VIF_ARGS);    \
            else vif_stat_wg_kernel<FW, OFF, ORDER_AVX, T><<<gw, VIF_THREADS, 0, s>>>(VIF_ARGS);       \
        } else if (order == ORDER_SCALAR) vif_stat_kernel<FW, OFF, ORDER_SCALAR, T><<<g, VIF_THREADS, 0, s>>>(VIF_ARGS);   \
        else if (order == ORDER_FMA) vif_stat_kernel<FW, OFF, ORDER_FMA, T><<<g, VIF_THREADS, 0, s>>>(VIF_ARGS);    \
        else vif_stat_avx_kernel<FW, OFF, T><<<gm, VIF_THREADS, 0, s>>>(VIF_ARGS, nb_pad);             \
    } while (0)
    switch (scale) {
    case 0:
        if (sample_bytes == 4) VIF_STAT(17, true, float);
        else if (sample_bytes == 1) VIF_STAT(17, true, unsigned char);
        else if (sample_bytes == 2) VIF_STAT(17, true, unsigned short);
        else return hipErrorInvalidValue;
        break;
    case 1: VIF_STAT(9, false, float); break;
    case 2: VIF_STAT(5, false, float); break;
    default: VIF_STAT(3, false, float); break;
    }
#undef VIF_STAT
#undef VIF_ARGS
    return hipGetLastError();
}

hipError_t vk_launch_vif_sums(const float *nd, int w, int h, int nframes, float *rowsum, double *out, size_t out_stride, int scale, hipStream_t s)
{
    vif_rowsum_kernel<<<dim3((h + 63) / 64, 2, nframes), 64, 0, s>>>(nd, w, h, rowsum);
    vif_total_kernel<<<dim3(2, nframes), 64, 0, s>>>(rowsum, h, out, out_stride, scale);
    return hipGetLastError();
}

hipError_t vk_launch_adm_scale(int scale, int order, const void *ref, const void *dis, int sample_bytes, size_t in_frame_stride, int in_row_stride,
                               int w, int h, float *ll_ref, float *ll_dis, size_t ll_frame_stride, int ll_row_stride,
                               float rfac_hv, float rfac_d, int left, int top, int right, int bottom, int nframes,
                               double *partial, hipStream_t s)
{
    const int bw = (w + 1) / 2, bh = (h + 1) / 2;
    const dim3 g = grid_flat(bw, bh, ADM_TW, ADM_TH, nframes);
    if (sample_bytes != 4 && scale != 0) return hipErrorInvalidValue;
#define ADM_ARGS (const T *)ref, (const T *)dis, in_frame_stride, in_row_stride, w, h, ll_ref, ll_dis, ll_frame_stride, ll_row_stride, rfac_hv, rfac_d, left, top, right, bottom, partial
#define ADM_SCALE(OFF, TYPE)                                                                                   \
    do {                                                                                                       \
        using T = TYPE;                                                                                        \
        if (order == ORDER_FMA) adm_scale_kernel<OFF, true, T><<<g, ADM_THREADS, 0, s>>>(ADM_ARGS);            \
        else adm_scale_kernel<OFF, false, T><<<g, ADM_THREADS, 0, s>>>(ADM_ARGS);                              \
    } while (0)
    if (scale != 0) ADM_SCALE(false, float);
    else if (sample_bytes == 4) ADM_SCALE(true, float);
    else if (sample_bytes == 1) ADM_SCALE(true, unsigned char);
    else if (sample_bytes == 2) ADM_SCALE(true, unsigned short);
    else return hipErrorInvalidValue;
#undef ADM_SCALE
#undef ADM_ARGS
    return hipGetLastError();
}

hipError_t vk_launch_motion(int order, const void *ref, int sample_bytes, size_t frame_stride_bytes, int row_stride_bytes, int w, int h,
                            int nframes, int has_carry, const float *carry_in, float *carry_out, double *partial, hipStream_t s)
{
    // wave = (band of 8 or 4 rows, chunk of frames), marching over the columns (motion_kernel): the picture's width is serial,
    // so the launch gets its parallelism from rows x chunks.  Every chunk pays one extra blur per step (of the frame before
    // it), so chunks are not made shorter than 4 frames; narrower bands come first.
    static const int forced = getenv("VMAF_HIP_MOTION_CHUNK") ? atoi(getenv("VMAF_HIP_MOTION_CHUNK")) : 0;
    static const int forced_bh = getenv("VMAF_HIP_MOTION_ROWS") ? atoi(getenv("VMAF_HIP_MOTION_ROWS")) : 0;
    // 8 rows per wave and chunks of whole chain groups (4 frames) when that still gives ~3000 waves (1080p x 120 frames: 4050);
    // otherwise 4 rows per wave (groups of 8 frames, chunks of 4)
    const int rows8 = (h + 7) / 8;
    int bh = (long long)rows8 * ((nframes + 3) / 4) >= 3000 ? 8 : 4;
    if (forced_bh == 4 || forced_bh == 8) bh = forced_bh;
    int chunk = forced > 0 ? forced : 4;
    if (chunk > MO_CMAX) chunk = MO_CMAX;
    if (chunk > nframes) chunk = nframes;
    const int nchunks = (nframes + chunk - 1) / chunk;
    const int nbands = (h + bh * MO_WAVES - 1) / (bh * MO_WAVES);
    const dim3 g(nbands, nchunks, 1);
    const int tiles = vk_motion_tiles(w, h);
    float *rowsum = reinterpret_cast<float *>(partial + 1);
    const size_t rs_stride = (size_t)tiles * 2;
#define MOTION_ARGS frame_stride_bytes, row_stride_bytes, w, h, nframes, chunk, has_carry, carry_in, carry_out, rowsum, rs_stride
#define MOTION_OB(O, T, B) motion_kernel<O, T, B><<<g, MO_THREADS, 0, s>>>((const T *)ref, MOTION_ARGS)
#define MOTION_O(O, T) do { if (bh == 8) MOTION_OB(O, T, 8); else MOTION_OB(O, T, 4); } while (0)
#define MOTION_T(T)                                           \
    if (order == ORDER_SCALAR) MOTION_O(ORDER_SCALAR, T);     \
    else if (order == ORDER_FMA) MOTION_O(ORDER_FMA, T);      \
    else MOTION_O(ORDER_AVX, T)
    if (sample_bytes == 4) { MOTION_T(float); }
    else if (sample_bytes == 1) { MOTION_T(unsigned char); }
    else if (sample_bytes == 2) { MOTION_T(unsigned short); }
    else return hipErrorInvalidValue;
#undef MOTION_T
#undef MOTION_O
#undef MOTION_OB
#undef MOTION_ARGS
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    motion_total_kernel<<<nframes, 64, 0, s>>>(rowsum, rs_stride, h, partial, (size_t)tiles);
    return hipGetLastError();
}

hipError_t vk_launch_widen(const void *src, int bytes_per_sample, size_t src_row_bytes, size_t src_frame_bytes, float *dst,
                           int w, int h, int nframes, hipStream_t s)
{
    const dim3 g((w + 1023) / 1024, h, nframes);
    const size_t a = (size_t)4 * bytes_per_sample;                    // alignment of a 4-sample vector load
    const int vec_ok = w % 4 == 0 && reinterpret_cast<uintptr_t>(src) % a == 0 && src_row_bytes % a == 0 && src_frame_bytes % a == 0 &&
                       reinterpret_cast<uintptr_t>(dst) % 16 == 0;
    if (bytes_per_sample == 1)
        widen_kernel<unsigned char><<<g, 256, 0, s>>>((const unsigned char *)src, src_row_bytes, src_frame_bytes, dst, w, h, vec_ok);
    else
        widen_kernel<unsigned short><<<g, 256, 0, s>>>((const unsigned char *)src, src_row_bytes, src_frame_bytes, dst, w, h, vec_ok);
    return hipGetLastError();
}

int vk_psnr_tiles(int h) { return (h + PSNR_ROWS - 1) / PSNR_ROWS; }
int vk_ssim_tiles(int w, int h) { return w < 11 || h < 11 ? 0 : ((w - 10 + SS_TW - 1) / SS_TW) * ((h - 10 + SS_TH - 1) / SS_TH); }

hipError_t vk_launch_psnr(const float *ref, const float *dis, size_t frame_stride, int row_stride, int w, int h, int nframes,
                          double *partial, hipStream_t s)
{
    psnr_kernel<<<dim3(vk_psnr_tiles(h), 1, nframes), 256, 0, s>>>(ref, dis, frame_stride, row_stride, w, h, partial);
    return hipGetLastError();
}

hipError_t vk_launch_box_decimate(const float *src, size_t src_frame_stride, int src_row_stride, int w, int h, int f, float *dst,
                                  int sw, int sh, int nframes, hipStream_t s)
{
    box_decimate_kernel<<<dim3((sw + 63) / 64, (sh + 3) / 4, nframes), 256, 0, s>>>(src, src_frame_stride, src_row_stride, w, h, f, dst, sw, sh);
    return hipGetLastError();
}

// both planes of every pair; f = 2 .. 8 through the coalesced kernel, larger factors (8K and up) through the plain one
hipError_t vk_launch_box_decimate_pair(const float *ref, const float *dis, size_t src_frame_stride, int src_row_stride, int w, int h, int f,
                                       float *oref, float *odis, int sw, int sh, int nframes, hipStream_t s)
{
    if (f > BOXR_FMAX) {
        hipError_t err = vk_launch_box_decimate(ref, src_frame_stride, src_row_stride, w, h, f, oref, sw, sh, nframes, s);
        return err != hipSuccess ? err : vk_launch_box_decimate(dis, src_frame_stride, src_row_stride, w, h, f, odis, sw, sh, nframes, s);
    }
    box_decimate_rows_kernel<<<dim3((sw + 63) / 64, (sh + 3) / 4, 2 * nframes), 256, 0, s>>>(ref, dis, src_frame_stride, src_row_stride, w, h, f,
                                                                                           oref, odis, sw, sh);
    return hipGetLastError();
}

hipError_t vk_launch_lpf_decimate(const float *ref, const float *dis, size_t src_frame_stride, int src_row_stride, int w, int h,
                                  float *oref, float *odis, int sw, int sh, int nframes, int exact, hipStream_t s)
{
    const dim3 g((sw + LPF_TW - 1) / LPF_TW, (sh + LPF_TH - 1) / LPF_TH, 2 * nframes);
    if (exact) lpf_decimate_kernel<true><<<g, 256, 0, s>>>(ref, dis, src_frame_stride, src_row_stride, w, h, oref, odis, sw, sh);
    else lpf_decimate_kernel<false><<<g, 256, 0, s>>>(ref, dis, src_frame_stride, src_row_stride, w, h, oref, odis, sw, sh);
    return hipGetLastError();
}

// ref, cmp: dense planes [nframes][h][w]; partial receives vk_ssim_tiles(w, h) * 4 doubles per frame
// psnr_partial: null, or vk_ssim_tiles(w, h) doubles per frame for the PSNR noise sums of the same planes (cs_only launches)
hipError_t vk_launch_ssim_stat(const float *ref, const float *cmp, int w, int h, int nframes, int cs_only, int exact, double *partial,
                               double *psnr_partial, hipStream_t s)
{
    const dim3 g((w - 10 + SS_TW - 1) / SS_TW, (h - 10 + SS_TH - 1) / SS_TH, nframes);
    if (psnr_partial && !cs_only) return hipErrorInvalidValue;      // the fused PSNR sum exists for MS-SSIM level 0 only
#define SSIM_STAT(EX)                                                                                             \
    if (psnr_partial) ssim_stat_kernel<true, true, EX><<<g, 256, 0, s>>>(ref, cmp, w, h, partial, psnr_partial);  \
    else if (cs_only) ssim_stat_kernel<true, false, EX><<<g, 256, 0, s>>>(ref, cmp, w, h, partial, nullptr);      \
    else ssim_stat_kernel<false, false, EX><<<g, 256, 0, s>>>(ref, cmp, w, h, partial, nullptr)
    if (exact) { SSIM_STAT(true); } else { SSIM_STAT(false); }
#undef SSIM_STAT
    return hipGetLastError();
}

int vk_ansnr_tiles(int w, int h) { return ((w + AN_TW - 1) / AN_TW) * ((h + AN_TH - 1) / AN_TH); }

// un-offset planes in (the kernel applies the -128 every caller of compute_ansnr applies); 2 doubles per workgroup
hipError_t vk_launch_ansnr(const float *ref, const float *dis, size_t frame_stride, int row_stride, int w, int h, int nframes,
                           double *partial, hipStream_t s)
{
    ansnr_kernel<<<dim3((w + AN_TW - 1) / AN_TW, (h + AN_TH - 1) / AN_TH, nframes), 256, 0, s>>>(ref, dis, frame_stride, row_stride, w, h, partial);
    return hipGetLastError();
}

hipError_t vk_launch_reduce(const double *partial, const vk_reduce_table *tab, int nframes, double *sums, hipStream_t s)
{
    reduce_kernel<<<dim3(VK_NSUMS, nframes), 256, 0, s>>>(partial, *tab, sums);
    return hipGetLastError();
}

hipError_t vk_launch_svm(const double *sv, const double *coef, int n_sv, int dim, double gamma, double rho, const double *x,
                         int n, double *out, hipStream_t s)
{
    svm_rbf_kernel<<<n, 256, 0, s>>>(sv, coef, n_sv, dim, gamma, rho, x, out);
    return hipGetLastError();
}

}  // extern "C"
