// Gather-type image operators of the MonoPSR instance path for gfx950: crop_and_resize, bilinear resize,
// max-pool and the root-convolution im2col.  All are HBM/L2-bound streaming kernels: one thread per output
// float4 (channels innermost, NHWC), fully coalesced stores, 16-byte loads when C % 4 == 0.
//
// Coordinate arithmetic is float32 in the order TensorFlow 1.8's kernels use (restated in oracle/net.py), and the
// file is built with -ffp-contract=off so source coordinates round exactly as there.
#include "common.h"

#pragma clang fp contract(off)

namespace {

template <int V>
struct Vec;
template <>
struct Vec<4> {
    using type = float4;
};
template <>
struct Vec<1> {
    using type = float;
};

__device__ __forceinline__ float4 lerp4(float4 a, float4 b, float t)
{
    return make_float4(a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t, a.z + (b.z - a.z) * t, a.w + (b.w - a.w) * t);
}
__device__ __forceinline__ float lerp4(float a, float b, float t) { return a + (b - a) * t; }
__device__ __forceinline__ float4 splat(float v, float4) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float splat(float v, float) { return v; }
__device__ __forceinline__ float4 vmax(float4 a, float4 b)
{
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }

// tf.image.crop_and_resize, bilinear.  Thread = one output pixel x V channels.
template <int V>
__global__ __launch_bounds__(256) void crop_and_resize_kernel(const float *__restrict__ image, int H, int W, int C,
                                                              const float *__restrict__ boxes,
                                                              const int *__restrict__ box_ind, int nimg, int ch,
                                                              int cw, float extrap, float *__restrict__ out,
                                                              long long total)
{
    using T = typename Vec<V>::type;
    const int cv = C / V;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        long long r = i / cv;
        const int x = (int)(r % cw);
        r /= cw;
        const int y = (int)(r % ch);
        const int bi = (int)(r / ch);
        const float y1 = boxes[4 * bi], x1 = boxes[4 * bi + 1], y2 = boxes[4 * bi + 2], x2 = boxes[4 * bi + 3];
        const int img = box_ind ? box_ind[bi] : 0;
        T res = splat(extrap, T());
        const float hs = ch > 1 ? (y2 - y1) * (float)(H - 1) / (float)(ch - 1) : 0.f;
        const float ws = cw > 1 ? (x2 - x1) * (float)(W - 1) / (float)(cw - 1) : 0.f;
        const float in_y = ch > 1 ? y1 * (float)(H - 1) + (float)y * hs : 0.5f * (y1 + y2) * (float)(H - 1);
        const float in_x = cw > 1 ? x1 * (float)(W - 1) + (float)x * ws : 0.5f * (x1 + x2) * (float)(W - 1);
        if (img >= 0 && img < nimg && !(in_y < 0.f) && !(in_y > (float)(H - 1)) && !(in_x < 0.f) &&
            !(in_x > (float)(W - 1)) && in_y == in_y && in_x == in_x) {
            const int top = (int)floorf(in_y), bot = (int)ceilf(in_y);
            const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
            const float yl = in_y - (float)top, xl = in_x - (float)left;
            const T *base = reinterpret_cast<const T *>(image + (size_t)img * H * W * C);
            const T tl = base[((size_t)top * W + left) * cv + c], tr = base[((size_t)top * W + right) * cv + c];
            const T bl = base[((size_t)bot * W + left) * cv + c], br = base[((size_t)bot * W + right) * cv + c];
            res = lerp4(lerp4(tl, tr, xl), lerp4(bl, br, xl), yl);
        }
        reinterpret_cast<T *>(out)[i] = res;
    }
}

// tf.image.resize_bilinear (TF 1.8 kernel).  Thread = one output pixel x V channels.
template <int V>
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float *__restrict__ in, int H, int W, int C,
                                                              int OH, int OW, float hscale, float wscale,
                                                              float *__restrict__ out, long long total)
{
    using T = typename Vec<V>::type;
    const int cv = C / V;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        long long r = i / cv;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const int b = (int)(r / OH);
        const float sy = (float)oy * hscale, sx = (float)ox * wscale;
        const int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float yl = sy - (float)y0, xl = sx - (float)x0;
        const T *base = reinterpret_cast<const T *>(in + (size_t)b * H * W * C);
        const T tl = base[((size_t)y0 * W + x0) * cv + c], tr = base[((size_t)y0 * W + x1) * cv + c];
        const T bl = base[((size_t)y1 * W + x0) * cv + c], br = base[((size_t)y1 * W + x1) * cv + c];
        reinterpret_cast<T *>(out)[i] = lerp4(lerp4(tl, tr, xl), lerp4(bl, br, xl), yl);
    }
}

// The same, one output row per blockIdx.y and one image per blockIdx.z: the index decode is one 32-bit division
// instead of three 64-bit ones (the generic kernel's decode was most of its instructions: 1.6 of a possible ~5 TB/s on
// the decoder's two 2x upsamplings).  Same arithmetic, same bits.
template <int V>
__global__ __launch_bounds__(256) void resize_bilinear_rows_kernel(const float *__restrict__ in, int H, int W, int C,
                                                                   int OH, int OW, float hscale, float wscale,
                                                                   float *__restrict__ out)
{
    using T = typename Vec<V>::type;
    const unsigned cv = (unsigned)(C / V);
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (unsigned)OW * cv) return;
    const unsigned ox = idx / cv, c = idx - ox * cv;
    const int oy = blockIdx.y, b = blockIdx.z;
    const float sy = (float)oy * hscale, sx = (float)ox * wscale;
    const int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float yl = sy - (float)y0, xl = sx - (float)x0;
    const T *base = reinterpret_cast<const T *>(in + (size_t)b * H * W * C);
    const T tl = base[((size_t)y0 * W + x0) * cv + c], tr = base[((size_t)y0 * W + x1) * cv + c];
    const T bl = base[((size_t)y1 * W + x0) * cv + c], br = base[((size_t)y1 * W + x1) * cv + c];
    reinterpret_cast<T *>(out)[((size_t)b * OH + oy) * OW * cv + idx] = lerp4(lerp4(tl, tr, xl), lerp4(bl, br, xl), yl);
}

// NHWC in, channel-blocked out ([B][C/8][OH][OW][8], the layout the decoder's Winograd kernels read: network.hip).
// Thread = one output pixel x 8 channels, lanes run along the output row: a wave writes 32-byte pieces that are
// contiguous in memory (the 4x larger side of the traffic); the reads are 32-byte pieces of the small, L2-resident
// source.  Same arithmetic as the kernels above.  grid (ceil(C/8 * OW / 256), OH, B).
__global__ __launch_bounds__(256) void resize_bilinear_c8out_kernel(const float *__restrict__ in, int H, int W, int C,
                                                                    int OH, int OW, float hscale, float wscale,
                                                                    float *__restrict__ out)
{
    const unsigned planes = (unsigned)C / 8u;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= planes * (unsigned)OW) return;
    const unsigned plane = idx / (unsigned)OW, ox = idx - plane * (unsigned)OW;
    const int oy = blockIdx.y, b = blockIdx.z;
    const float sy = (float)oy * hscale, sx = (float)ox * wscale;
    const int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float yl = sy - (float)y0, xl = sx - (float)x0;
    const float4 *base = reinterpret_cast<const float4 *>(in + (size_t)b * H * W * C) + plane * 2;
    const size_t cv = (size_t)C / 4;
    float4 *dst = reinterpret_cast<float4 *>(out + ((((size_t)b * planes + plane) * OH + oy) * OW + ox) * 8);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float4 tl = base[((size_t)y0 * W + x0) * cv + h], tr = base[((size_t)y0 * W + x1) * cv + h];
        const float4 bl = base[((size_t)y1 * W + x0) * cv + h], br = base[((size_t)y1 * W + x1) * cv + h];
        dst[h] = lerp4(lerp4(tl, tr, xl), lerp4(bl, br, xl), yl);
    }
}

// The same through LDS: the kernel above reads four 32-byte taps from L2 for every 32 bytes it writes -- on the decoder's
// two upsamplings 3.6 GB of L2 reads for 0.9 GB of output, and that, not the stores, sets its time.  Here a workgroup
// owns RO output rows of one channel block of one image, stages the few source rows they reach (8-channel pieces,
// 32 bytes per pixel) in LDS once, and every thread takes its four taps from there.  Same taps, same arithmetic:
// bit-identical results.  grid (C / 8, ceil(OH / RO), B), RO = 4 x (256 / OW) output rows in four passes; at most
// kResizeLdsMaxRows source rows per workgroup.
constexpr int kResizeLdsMaxRows = 16, kResizeLdsIters = 4;  // a workgroup writes 4 x (256 / OW) output rows
__global__ __launch_bounds__(256) void resize_bilinear_c8out_lds_kernel(const float *__restrict__ in, int H, int W, int C,
                                                                        int OH, int OW, int RO, float hscale,
                                                                        float wscale, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float4 rows[];  // [source row][x][2 pieces]
    const int plane = blockIdx.x, oy0 = blockIdx.y * RO, b = blockIdx.z;
    const int planes = C / 8;
    const int oy_last = min(oy0 + RO, OH) - 1;
    const int y_lo = (int)floorf((float)oy0 * hscale);
    const int y_hi = min((int)floorf((float)oy_last * hscale) + 1, H - 1);
    const int nrows = y_hi - y_lo + 1;
    const float4 *base = reinterpret_cast<const float4 *>(in + (size_t)b * H * W * C) + plane * 2;
    const size_t cv = (size_t)C / 4;
    for (int i = threadIdx.x; i < nrows * W * 2; i += 256) {
        const int h = i & 1, px = i >> 1;
        const int r = px / W, x = px - r * W;
        rows[i] = base[((size_t)(y_lo + r) * W + x) * cv + h];
    }
    __syncthreads();
    const int rpi = 256 / OW;  // output rows per pass of the workgroup
    const int r = threadIdx.x / OW, ox = threadIdx.x - r * OW;
    if (r >= rpi) return;
    const float sx = (float)ox * wscale;
    const int x0 = (int)floorf(sx), x1 = min(x0 + 1, W - 1);
    const float xl = sx - (float)x0;
    for (int oy = oy0 + r; oy <= oy_last; oy += rpi) {
        const float sy = (float)oy * hscale;
        const int y0 = (int)floorf(sy), y1 = min(y0 + 1, H - 1);
        const float yl = sy - (float)y0;
        float4 *dst = reinterpret_cast<float4 *>(out + ((((size_t)b * planes + plane) * OH + oy) * OW + ox) * 8);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 tl = rows[((y0 - y_lo) * W + x0) * 2 + h], tr = rows[((y0 - y_lo) * W + x1) * 2 + h];
            const float4 bl = rows[((y1 - y_lo) * W + x0) * 2 + h], br = rows[((y1 - y_lo) * W + x1) * 2 + h];
            dst[h] = lerp4(lerp4(tl, tr, xl), lerp4(bl, br, xl), yl);
        }
    }
}

// slim.max_pool2d; pad_top/pad_left are the SAME-padding offsets (0 for VALID); padded cells never win.
template <int V>
__global__ __launch_bounds__(256) void max_pool_kernel(const float *__restrict__ in, int H, int W, int C, int OH,
                                                       int OW, int k, int s, int pad_top, int pad_left,
                                                       float *__restrict__ out, long long total)
{
    using T = typename Vec<V>::type;
    const int cv = C / V;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        long long r = i / cv;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const int b = (int)(r / OH);
        const T *base = reinterpret_cast<const T *>(in + (size_t)b * H * W * C);
        T best = splat(-__builtin_inff(), T());
        for (int ky = 0; ky < k; ++ky) {
            const int y = oy * s - pad_top + ky;
            if (y < 0 || y >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int x = ox * s - pad_left + kx;
                if (x < 0 || x >= W) continue;
                best = vmax(best, base[((size_t)y * W + x) * cv + c]);
            }
        }
        reinterpret_cast<T *>(out)[i] = best;
    }
}

// Root im2col: explicit pad 3, 7x7 window, stride 2; row = output pixel, column (ky*7+kx)*3+c, zero tail to kpad.
__global__ __launch_bounds__(256) void im2col_root_kernel(const float *__restrict__ x, int H, int W, int OH, int OW,
                                                          int kpad, float *__restrict__ cols, long long total)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i % kpad);
        long long r = i / kpad;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const int b = (int)(r / OH);
        float v = 0.f;
        if (j < 147) {
            const int c = j % 3, t = j / 3;
            const int kx = t % 7, ky = t / 7;
            const int yy = oy * 2 + ky - 3, xx = ox * 2 + kx - 3;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = x[(((size_t)b * H + yy) * W + xx) * 3 + c];
        }
        cols[i] = v;
    }
}

// The same rows, four columns per thread (one 16-byte store), one output row per blockIdx.y, one image per
// blockIdx.z: 32-bit index arithmetic only.  kpad % 4 == 0.
__global__ __launch_bounds__(256) void im2col_root_rows_kernel(const float *__restrict__ x, int H, int W, int OW,
                                                               int kpad, float *__restrict__ cols)
{
    const unsigned q4 = (unsigned)kpad / 4u;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (unsigned)OW * q4) return;
    const unsigned ox = idx / q4, jq = idx - ox * q4;
    const int oy = blockIdx.y, b = blockIdx.z, OH = gridDim.y;
    const float *xb = x + (size_t)b * H * W * 3;
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = (int)jq * 4 + u;
        v[u] = 0.f;
        if (j < 147) {
            const int t = j / 3, c = j - 3 * t;
            const int ky = t / 7, kx = t - 7 * ky;
            const int yy = oy * 2 + ky - 3, xx = (int)ox * 2 + kx - 3;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) v[u] = xb[((size_t)yy * W + xx) * 3 + c];
        }
    }
    reinterpret_cast<float4 *>(cols)[((size_t)b * OH + oy) * OW * q4 + idx] = make_float4(v[0], v[1], v[2], v[3]);
}

inline int grid_for(long long total) { return (int)((total + 255) / 256 < 262144 ? (total + 255) / 256 : 262144); }

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int mpsr_crop_and_resize(const float *image, int nimg, int H, int W, int C, const float *boxes,
                                    const int *box_ind, int nb, int ch, int cw, float extrapolation_value, float *out,
                                    mpsr_stream_t stream)
{
    MPSR_REQUIRE(nimg > 0 && H > 0 && W > 0 && C > 0 && nb >= 0 && ch > 0 && cw > 0,
                 "crop_and_resize: bad shape (nimg=%d H=%d W=%d C=%d nb=%d crop=%dx%d)", nimg, H, W, C, nb, ch, cw);
    if (nb == 0) return MPSR_OK;
    MPSR_REQUIRE(image && boxes && out, "crop_and_resize: null pointer");
    hipStream_t s = mpsr::as_stream(stream);
    if (C % 4 == 0 && aligned16(image) && aligned16(out)) {
        const long long total = (long long)nb * ch * cw * (C / 4);
        hipLaunchKernelGGL(crop_and_resize_kernel<4>, dim3(grid_for(total)), dim3(256), 0, s, image, H, W, C, boxes,
                           box_ind, nimg, ch, cw, extrapolation_value, out, total);
    } else {
        const long long total = (long long)nb * ch * cw * C;
        hipLaunchKernelGGL(crop_and_resize_kernel<1>, dim3(grid_for(total)), dim3(256), 0, s, image, H, W, C, boxes,
                           box_ind, nimg, ch, cw, extrapolation_value, out, total);
    }
    MPSR_CHECK_LAUNCH("crop_and_resize_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_resize_bilinear(const float *in, int B, int H, int W, int C, int OH, int OW, int align_corners,
                                    float *out, mpsr_stream_t stream)
{
    MPSR_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0, "resize_bilinear: bad shape");
    if (B == 0) return MPSR_OK;
    MPSR_REQUIRE(in && out, "resize_bilinear: null pointer");
    const float hscale = (align_corners && OH > 1) ? (float)(H - 1) / (float)(OH - 1) : (float)H / (float)OH;
    const float wscale = (align_corners && OW > 1) ? (float)(W - 1) / (float)(OW - 1) : (float)W / (float)OW;
    hipStream_t s = mpsr::as_stream(stream);
    if (C % 4 == 0 && aligned16(in) && aligned16(out) && B <= 65535 && OH <= 65535 &&
        (long long)OW * (C / 4) < 0x7fffffffLL) {
        const dim3 grid((unsigned)mpsr::ceil_div(OW * (C / 4), 256), (unsigned)OH, (unsigned)B);
        hipLaunchKernelGGL(resize_bilinear_rows_kernel<4>, grid, dim3(256), 0, s, in, H, W, C, OH, OW, hscale, wscale,
                           out);
    } else if (C % 4 == 0 && aligned16(in) && aligned16(out)) {
        const long long total = (long long)B * OH * OW * (C / 4);
        hipLaunchKernelGGL(resize_bilinear_kernel<4>, dim3(grid_for(total)), dim3(256), 0, s, in, H, W, C, OH, OW,
                           hscale, wscale, out, total);
    } else {
        const long long total = (long long)B * OH * OW * C;
        hipLaunchKernelGGL(resize_bilinear_kernel<1>, dim3(grid_for(total)), dim3(256), 0, s, in, H, W, C, OH, OW,
                           hscale, wscale, out, total);
    }
    MPSR_CHECK_LAUNCH("resize_bilinear_kernel");
    return MPSR_OK;
}

namespace mpsr {
// tf.image.resize_bilinear with the output written channel-blocked (see resize_bilinear_c8out_kernel); C % 8 == 0.
int resize_bilinear_c8(const float *in, int B, int H, int W, int C, int OH, int OW, int align_corners, float *out,
                       hipStream_t s)
{
    MPSR_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && OH > 0 && OW > 0 && B <= 65535 && OH <= 65535 && in &&
                     out && aligned16(in) && aligned16(out),
                 "resize_bilinear_c8: bad arguments");
    const float hscale = (align_corners && OH > 1) ? (float)(H - 1) / (float)(OH - 1) : (float)H / (float)OH;
    const float wscale = (align_corners && OW > 1) ? (float)(W - 1) / (float)(OW - 1) : (float)W / (float)OW;
    // a workgroup per (channel block, RO output rows, image) with the source rows in LDS when the rows are short enough
    // for one workgroup to cover several of them and they reach few source rows
    if (OW <= 128 && C / 8 <= 65535) {
        // as many passes (up to four) as keep the source rows a workgroup reaches within the LDS window
        int RO = 0, span = 0;
        // (measured on the decoder's two: 48 output rows 217 -> 191 us with four passes, 24 rows 84 -> 93 us with three)
        for (int it = OH >= 40 ? kResizeLdsIters : 1; it >= 1; --it) {
            RO = it * (256 / OW);
            if (RO > OH) RO = OH;
            span = (int)floorf((float)(RO - 1) * hscale) + 3;  // source rows RO output rows can reach (bound)
            if (span <= kResizeLdsMaxRows) break;
        }
        if (span <= kResizeLdsMaxRows) {
            const dim3 grid((unsigned)(C / 8), (unsigned)ceil_div(OH, RO), (unsigned)B);
            const size_t lds = (size_t)kResizeLdsMaxRows * W * 2 * sizeof(float4);
            if (lds <= 48 * 1024) {
                hipLaunchKernelGGL(resize_bilinear_c8out_lds_kernel, grid, dim3(256), lds, s, in, H, W, C, OH, OW, RO,
                                   hscale, wscale, out);
                MPSR_CHECK_LAUNCH("resize_bilinear_c8out_lds_kernel");
                return MPSR_OK;
            }
        }
    }
    const dim3 grid((unsigned)ceil_div((C / 8) * OW, 256), (unsigned)OH, (unsigned)B);
    hipLaunchKernelGGL(resize_bilinear_c8out_kernel, grid, dim3(256), 0, s, in, H, W, C, OH, OW, hscale, wscale, out);
    MPSR_CHECK_LAUNCH("resize_bilinear_c8out_kernel");
    return MPSR_OK;
}
}  // namespace mpsr

extern "C" int mpsr_max_pool(const float *in, int B, int H, int W, int C, int k, int s_, int pad_same, float *out,
                             mpsr_stream_t stream)
{
    MPSR_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && k > 0 && s_ > 0, "max_pool: bad shape");
    if (B == 0) return MPSR_OK;
    MPSR_REQUIRE(in && out, "max_pool: null pointer");
    int OH, OW, pt = 0, pl = 0;
    if (pad_same) {
        OH = mpsr::ceil_div(H, s_);
        OW = mpsr::ceil_div(W, s_);
        const int th = (OH - 1) * s_ + k - H, tw = (OW - 1) * s_ + k - W;
        pt = (th > 0 ? th : 0) / 2;
        pl = (tw > 0 ? tw : 0) / 2;
    } else {
        MPSR_REQUIRE(H >= k && W >= k, "max_pool: VALID window %d larger than input %dx%d", k, H, W);
        OH = (H - k) / s_ + 1;
        OW = (W - k) / s_ + 1;
    }
    hipStream_t s = mpsr::as_stream(stream);
    if (C % 4 == 0 && aligned16(in) && aligned16(out)) {
        const long long total = (long long)B * OH * OW * (C / 4);
        hipLaunchKernelGGL(max_pool_kernel<4>, dim3(grid_for(total)), dim3(256), 0, s, in, H, W, C, OH, OW, k, s_, pt,
                           pl, out, total);
    } else {
        const long long total = (long long)B * OH * OW * C;
        hipLaunchKernelGGL(max_pool_kernel<1>, dim3(grid_for(total)), dim3(256), 0, s, in, H, W, C, OH, OW, k, s_, pt,
                           pl, out, total);
    }
    MPSR_CHECK_LAUNCH("max_pool_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_im2col_root(const float *x, int B, int H, int W, float *cols, int kpad, mpsr_stream_t stream)
{
    MPSR_REQUIRE(B >= 0 && H >= 1 && W >= 1, "im2col_root: bad shape");
    MPSR_REQUIRE(kpad >= 147 && kpad % 32 == 0, "im2col_root: kpad=%d must be a multiple of 32 and >= 147", kpad);
    if (B == 0) return MPSR_OK;
    MPSR_REQUIRE(x && cols, "im2col_root: null pointer");
    const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
    if (B <= 65535 && OH <= 65535 && aligned16(cols)) {
        const dim3 grid((unsigned)mpsr::ceil_div(OW * (kpad / 4), 256), (unsigned)OH, (unsigned)B);
        hipLaunchKernelGGL(im2col_root_rows_kernel, grid, dim3(256), 0, mpsr::as_stream(stream), x, H, W, OW, kpad,
                           cols);
    } else {
        const long long total = (long long)B * OH * OW * kpad;
        hipLaunchKernelGGL(im2col_root_kernel, dim3(grid_for(total)), dim3(256), 0, mpsr::as_stream(stream), x, H, W,
                           OH, OW, kpad, cols, total);
    }
    MPSR_CHECK_LAUNCH("im2col_root_kernel");
    return MPSR_OK;
}

// ------------------------------------------------------------------------------------------------ narrow 3x3 conv
//
// 3x3 stride-1 SAME convolution with a tiny output width (N <= 4): the xyz-map head (128 -> 3,
// monopsr_output_builder.py:95-104).  On the MFMA GEMM an N = 3 layer wastes 29/32 of every tile, and the layer is
// bound by reading its (B,48,48,128) input once (302 MB at B = 256), so it is a direct VALU kernel instead:
// a workgroup owns a 16x16 output tile, stages the 18x18 halo of 32 channels at a time in LDS (row stride 36 floats:
// conflict-free 16-byte reads), one thread per output pixel, weights read through the scalar cache (wave-uniform).
namespace {

constexpr int kNarrowTile = 16, kNarrowHalo = kNarrowTile + 2, kNarrowCh = 32, kNarrowStride = kNarrowCh + 4;

template <int NOUT>
__global__ __launch_bounds__(256) void conv3x3_narrow_kernel(const float *__restrict__ x, int H, int W, int C,
                                                             const float *__restrict__ w,
                                                             const float *__restrict__ bias, int relu,
                                                             float *__restrict__ y)
{
    __shared__ __attribute__((aligned(16))) float tile[kNarrowHalo * kNarrowHalo * kNarrowStride];
    const int b = blockIdx.z, ty0 = blockIdx.y * kNarrowTile, tx0 = blockIdx.x * kNarrowTile;
    const int tid = threadIdx.x, ly = tid >> 4, lx = tid & 15;
    const float *xb = x + (size_t)b * H * W * C;
    float acc[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) acc[o] = 0.f;
    const int K = 9 * C;
    for (int c0 = 0; c0 < C; c0 += kNarrowCh) {
        __syncthreads();
        for (int i = tid; i < kNarrowHalo * kNarrowHalo * (kNarrowCh / 4); i += 256) {
            const int pix = i >> 3, q = i & 7;
            const int py = pix / kNarrowHalo, px = pix - py * kNarrowHalo;
            const int gy = ty0 + py - 1, gx = tx0 + px - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W && c0 + q * 4 < C)
                v = *reinterpret_cast<const float4 *>(xb + ((size_t)gy * W + gx) * C + c0 + q * 4);
            *reinterpret_cast<float4 *>(&tile[pix * kNarrowStride + q * 4]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float *src = &tile[((ly + t / 3) * kNarrowHalo + lx + t % 3) * kNarrowStride];
            const float *wt = w + t * C + c0;  // + o * K ; wave-uniform -> scalar loads
#pragma unroll
            for (int q = 0; q < kNarrowCh / 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(src + q * 4);
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    const float *wo = wt + (size_t)o * K + q * 4;
                    acc[o] = fmaf(v.x, wo[0], acc[o]);
                    acc[o] = fmaf(v.y, wo[1], acc[o]);
                    acc[o] = fmaf(v.z, wo[2], acc[o]);
                    acc[o] = fmaf(v.w, wo[3], acc[o]);
                }
            }
        }
    }
    const int oy = ty0 + ly, ox = tx0 + lx;
    if (oy < H && ox < W) {
        float *dst = y + (((size_t)b * H + oy) * W + ox) * NOUT;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float v = acc[o] + (bias ? bias[o] : 0.f);
            dst[o] = relu ? fmaxf(v, 0.f) : v;
        }
    }
}

// The same layer on the matrix pipes, for N <= 3 (the xyz head itself): taps go into the GEMM's N.  With
//     G[p][3 t + o] = sum_c x[p][c] * w[o][t][c]            (a 1x1 GEMM: K = C, N = 9 taps x 3 outputs = 27 <= 32)
// the convolution is y[p][o] = bias[o] + sum_t G[p + delta_t][3 t + o]: a 27-wide v_mfma_f32_32x32x2_f32 tile wastes
// 5/32 instead of 29/32, the input is read ONCE (no 18x18 halo re-staging: a workgroup takes R + 2 full-width image
// rows, 1.25x at R = 8), and the tap sum is a 9-term LDS gather.  The weights live in registers as B fragments for the
// whole kernel (16 x 16 bytes per lane at C = 128); A fragments come straight from global memory in fragment layout
// (lane = pixel row x k half, 16 bytes: 32-byte runs per pixel, both halves of a 64-byte line by consecutive
// requests) -- nothing but G goes through LDS.  grid (ceil(H / R), B), 256 threads, dynamic LDS (R+2) * W * 27 floats.
constexpr int kNarrowRows = 8;
using nf32x16 = __attribute__((ext_vector_type(16))) float;

template <int KB, bool IN_C8>  // K blocks of 8 channels held as B fragments (C = 8 * KB); input NHWC or [C/8][H][W][8]
__global__ __launch_bounds__(256) void conv3x3_narrow_mfma_kernel(const float *__restrict__ x, int H, int W,
                                                                  const float *__restrict__ w,
                                                                  const float *__restrict__ bias, int relu, int nout,
                                                                  float *__restrict__ y, unsigned xbytes)
{
    extern __shared__ __attribute__((aligned(16))) float g[];  // [(R+2) * W pixels][27]
    constexpr int C = 8 * KB;
    const int b = blockIdx.y, y0 = blockIdx.x * kNarrowRows;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = (kNarrowRows + 2) * W;  // pixels of rows y0 - 1 .. y0 + R, row-major
    // B fragments: lane (n = lane & 31, k half = lane >> 5) holds W'[n][8 j + 4 h .. + 3], W'[3 t + o][c] = w[o][t][c]
    float4 bf[KB];
    {
        const int n = lane & 31, t = n / 3, o = n - 3 * t;
        const bool live = n < 9 * 3 && o < nout;
#pragma unroll
        for (int j = 0; j < KB; ++j)
            bf[j] = live ? *reinterpret_cast<const float4 *>(w + (size_t)o * 9 * C + (size_t)t * C + 8 * j + 4 * (lane >> 5))
                         : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)xbytes, 0x00020000);
    const int ntiles = (P + 31) / 32;
    for (int rt = wave; rt < ntiles; rt += 4) {
        // this lane's pixel: p = 32 rt + (lane & 31) -> image row y0 - 1 + p / W; rows outside the image and pixels past
        // P read as zeros (out-of-range offset), so their G is exactly 0
        const int pix = 32 * rt + (lane & 31);
        const int py = pix / W, gy = y0 - 1 + py;
        const bool ok = pix < P && gy >= 0 && gy < H;
        // (C8: the 32 pixels of a tile are 32 x 32 contiguous bytes per K block -- one request = 1 KiB contiguous)
        const unsigned off = !ok ? 0x80000000u
                             : IN_C8 ? (unsigned)((((size_t)b * KB * H + gy) * W + (pix - py * W)) * 8 + 4 * (lane >> 5)) * 4u
                                     : (unsigned)((((size_t)b * H + gy) * W + (pix - py * W)) * C + 4 * (lane >> 5)) * 4u;
        const unsigned kstride = IN_C8 ? (unsigned)H * (unsigned)W * 32u : 32u;  // bytes between K blocks
        nf32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        float4 a[KB];
#pragma unroll
        for (int j = 0; j < KB; ++j)
            a[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, (unsigned)j * kstride, 0));
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].x, bf[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].y, bf[j].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].z, bf[j].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].w, bf[j].w, acc, 0, 0, 0);
        }
        // accumulator element e of a lane: pixel row (e & 3) + 8 (e >> 2) + 4 (lane >> 5) of the tile, column n = lane & 31
        const int n = lane & 31;
        if (n < 27) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int pr = 32 * rt + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (pr < P) g[pr * 27 + n] = acc[e];
            }
        }
    }
    __syncthreads();
    // tap sum: output pixel (y0 + r, xx) reads G of tile row r + ky (ky = 0..2), column xx + kx - 1
    const int nrows = min(kNarrowRows, H - y0);
    for (int i = tid; i < nrows * W; i += 256) {
        const int r = i / W, xx = i - r * W;
        float out[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int sx = xx + kx - 1;
                if (sx < 0 || sx >= W) continue;
                const float *src = g + ((r + ky) * W + sx) * 27 + 3 * (ky * 3 + kx);
#pragma unroll
                for (int o = 0; o < 3; ++o) out[o] += src[o];
            }
        float *dst = y + (((size_t)b * H + y0 + r) * W + xx) * nout;
        for (int o = 0; o < nout; ++o) {
            const float v = out[o] + (bias ? bias[o] : 0.f);
            dst[o] = relu ? fmaxf(v, 0.f) : v;
        }
    }
}

}  // namespace

namespace mpsr {
// Used by conv2d() for 3x3, dilation 1, N <= 4, C % 32 == 0 layers without residual.
// True when conv3x3_narrow runs the taps-in-N MFMA kernel for this layer -- the only form that reads a channel-blocked
// input.  ONE copy of the rule: mpsr_squash_decoder_fwd (network.hip) asks it before it lets the last decoder layer
// write channel-blocked (x = nullptr: the input does not exist yet, its alignment is the workspace's 256 bytes).
bool conv3x3_narrow_takes_mfma(const float *x, int B, int H, int W, int C, int N, const float *w)
{
    const size_t lds = (size_t)(kNarrowRows + 2) * W * 27 * sizeof(float);
    const long long xb = (long long)B * H * W * C * 4;
    const bool inst = C == 32 || C == 64 || C == 96 || C == 128;  // the instantiated channel counts
    return N >= 1 && N <= 3 && inst && lds <= 64 * 1024 && B <= 65535 && xb < 0x7fffffffLL && ((uintptr_t)x & 15) == 0 &&
           ((uintptr_t)w & 15) == 0;
}

int conv3x3_narrow(const float *x, int B, int H, int W, int C, const float *w, const float *bias, int relu, float *y,
                   int N, hipStream_t s, int in_c8)
{
    // N <= 3 with C = 32 .. 128: the GEMM-with-taps-in-N kernel on the matrix pipes (above)
    {
        const size_t lds = (size_t)(kNarrowRows + 2) * W * 27 * sizeof(float);
        const long long xb = (long long)B * H * W * C * 4;
        if (conv3x3_narrow_takes_mfma(x, B, H, W, C, N, w)) {
            const dim3 grid(ceil_div(H, kNarrowRows), B);
#define MPSR_NARROW(KB_)                                                                                              \
    if (in_c8) {                                                                                                      \
        MPSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_narrow_mfma_kernel<KB_, true>),       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                     \
        hipLaunchKernelGGL((conv3x3_narrow_mfma_kernel<KB_, true>), grid, dim3(256), lds, s, x, H, W, w, bias, relu,   \
                           N, y, (unsigned)xb);                                                                        \
    } else {                                                                                                          \
        MPSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_narrow_mfma_kernel<KB_, false>),      \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                     \
        hipLaunchKernelGGL((conv3x3_narrow_mfma_kernel<KB_, false>), grid, dim3(256), lds, s, x, H, W, w, bias, relu,  \
                           N, y, (unsigned)xb);                                                                        \
    }
            switch (C / 8) {
                case 4: MPSR_NARROW(4); break;
                case 8: MPSR_NARROW(8); break;
                case 12: MPSR_NARROW(12); break;
                case 16: MPSR_NARROW(16); break;
                default: goto direct;
            }
#undef MPSR_NARROW
            MPSR_CHECK_LAUNCH("conv3x3_narrow_mfma_kernel");
            return MPSR_OK;
        }
    }
direct:
    if (in_c8)
        return fail(MPSR_ERR_UNSUPPORTED, "conv3x3_narrow: a channel-blocked input needs N <= 3 and C = 32, 64, 96 or 128");
    dim3 grid(ceil_div(W, kNarrowTile), ceil_div(H, kNarrowTile), B);
    if (grid.z > 65535) return fail(MPSR_ERR_UNSUPPORTED, "conv3x3_narrow: batch %d exceeds 65535", B);
    switch (N) {
        case 1: hipLaunchKernelGGL(conv3x3_narrow_kernel<1>, grid, dim3(256), 0, s, x, H, W, C, w, bias, relu, y); break;
        case 2: hipLaunchKernelGGL(conv3x3_narrow_kernel<2>, grid, dim3(256), 0, s, x, H, W, C, w, bias, relu, y); break;
        case 3: hipLaunchKernelGGL(conv3x3_narrow_kernel<3>, grid, dim3(256), 0, s, x, H, W, C, w, bias, relu, y); break;
        case 4: hipLaunchKernelGGL(conv3x3_narrow_kernel<4>, grid, dim3(256), 0, s, x, H, W, C, w, bias, relu, y); break;
        default: return fail(MPSR_ERR_UNSUPPORTED, "conv3x3_narrow: N=%d", N);
    }
    MPSR_CHECK_LAUNCH("conv3x3_narrow_kernel");
    return MPSR_OK;
}
}  // namespace mpsr
