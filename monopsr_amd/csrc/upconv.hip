// 3x3 SAME convolution of a bilinearly UPSAMPLED map without ever forming the upsampled map: the map decoder's
// conv2_1 and conv3_1 (reference graph monopsr/builders/net_builder.py:72-77 and :81-85 -- tf.image.resize_bilinear
// (align_corners) straight into slim.conv2d 3x3).
//
// Channel mixing commutes with a per-channel spatial operator.  With U = the bilinear upsampling (out pixel q reads four
// source pixels s with weights a[q][s] that depend on q only), tap t = (dy, dx) and g = the (N, 3, 3, C) filter:
//     y[p][n] = bias[n] + sum_t [p + t inside the upsampled image] sum_c g[n][t][c] U(x)[p + t][c]
//             = bias[n] + sum_t [p + t inside]                      sum_s a[p + t][s] z[s][t][n],
//     z[s][t][n] = sum_c g[n][t][c] x[s][c]                 -- a 1x1 convolution of the SOURCE map with 9 N outputs.
// Stage 1 is therefore a plain GEMM [source pixels x C] . [C x 9 N] on the persistent pointwise kernel (pointwise.hip):
// the same 1/4 of the direct form's multiply-adds that F(4x4,3x3) on the upsampled map issues (a quarter of the pixels,
// nine times the columns), but with no input / output transforms, no patch gathers, exact-fp32 GEMM error (1e-6 instead
// of F(4x4)'s 1.5e-5), and the two resize launches (0.9 GB written and re-read per step) are gone.  Stage 2
// (upconv_gather_kernel) is the 9-tap x 4-corner weighted sum above: one workgroup per (image, band of output rows,
// 8-channel block) stages the z rows its band reaches in LDS once and every thread then sums 36 float4 for an output
// pixel's 4 channels -- LDS / vector-ALU work on 1/9 of the GEMM's flops, writing the layer's output once (channel-
// blocked for the F(4x4,3x3) layer that follows, or NHWC).
//
// Column order of z (ours to choose): j = (n / 8) * 72 + t * 8 + (n % 8) -- the nine taps of an 8-channel block are 288
// contiguous bytes per source pixel.  The GEMM's weight matrix W'[j][c] = g[n][t][c] is a re-ordering of the filter
// rows (upconv_weights_kernel; kept in the caller's filter cache when there is one).  9 N columns are cut into parts of
// 1152 = 9 column blocks of the pointwise kernel (its persistent grid then fills the chip: 504 of 512 workgroup slots),
// i.e. one part per 128 output channels.
#include "common.h"
#include "wino3_filter.h"

namespace mpsr {
bool pointwise_applies(long long M, int K, int N);
int conv1x1_pointwise(const float *x, long long M, int K, const float *w, const float *bias, const float *residual,
                      int relu, float *y, int N, hipStream_t s);
}  // namespace mpsr

namespace {

constexpr int PART = 1152;  // z columns per GEMM launch = 16 channel blocks x 9 taps x 8 channels
constexpr size_t kGatherLdsBytes = 80 * 1024;

// W'[j][c] = g[n][t*C + c], j = (n / 8) * 72 + t * 8 + (n % 8).  One thread per float4 of W'.
__global__ __launch_bounds__(256) void upconv_weights_kernel(const float *__restrict__ g, int N, int C, float *__restrict__ wp)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C / 4;
    if (i >= (long long)9 * N * c4) return;
    const int j = (int)(i / c4), q = (int)(i - (long long)j * c4);
    const int blk = j / 72, r = j - blk * 72, t = r >> 3, n = blk * 8 + (r & 7);
    reinterpret_cast<float4 *>(wp)[i] = reinterpret_cast<const float4 *>(g + ((size_t)n * 9 + t) * C)[q];
}

struct UpcParams {
    const float *z, *bias;
    float *y;
    size_t part_stride;  // floats between the z parts (source pixels x 1152)
    int h, w, H, W, N, relu, RB;
    float hscale, wscale;
};

// grid (N / 8, bands of RB output rows, images).  LDS: [source row][source column][tap][8 channels] as float4 pairs.
template <bool OUT_C8>
__global__ __launch_bounds__(256) void upconv_gather_kernel(const UpcParams p)
{
    extern __shared__ __attribute__((aligned(16))) float4 src4[];
    const int tid = threadIdx.x;
    const int blk = blockIdx.x, y0 = blockIdx.y * p.RB, b = blockIdx.z;
    const int ylast = min(y0 + p.RB, p.H) - 1;
    // source rows the band's taps (output rows y0 - 1 .. ylast + 1, clamped) interpolate between
    const int r_lo = (int)floorf((float)max(y0 - 1, 0) * p.hscale);
    const int r_hi = min((int)floorf((float)min(ylast + 1, p.H - 1) * p.hscale) + 1, p.h - 1);
    const int nrows = r_hi - r_lo + 1;
    {
        const float *zp = p.z + (size_t)(blk >> 4) * p.part_stride + (size_t)(blk & 15) * 72;
        const size_t pix0 = ((size_t)b * p.h + r_lo) * p.w;
        const int total4 = nrows * p.w * 18;
        for (int i = tid; i < total4; i += 256) {
            const int px = i / 18, q = i - px * 18;
            src4[i] = *reinterpret_cast<const float4 *>(zp + (pix0 + px) * PART + 4 * q);
        }
    }
    __syncthreads();
    const int nitems = (ylast - y0 + 1) * p.W * 2;  // (output pixel, half of the 8 channels)
    const int n0 = blk * 8;
    for (int it = tid; it < nitems; it += 256) {
        const int half = it & 1, px = it >> 1;
        const int yy = px / p.W, x = px - yy * p.W, y = y0 + yy;
        // tap row / column d - 1: the two source rows (columns) it interpolates between, as LDS offsets, and their
        // weights -- zero when the tap falls outside the upsampled image (the convolution's SAME padding)
        int ro[3][2], co[3][2];
        float wy[3][2], wx[3][2];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int qy = y + d - 1, qx = x + d - 1;
            const bool vy = qy >= 0 && qy < p.H, vx = qx >= 0 && qx < p.W;
            const float sy = (float)min(max(qy, 0), p.H - 1) * p.hscale, sx = (float)min(max(qx, 0), p.W - 1) * p.wscale;
            const int r0 = (int)floorf(sy), c0 = (int)floorf(sx);
            const int r1 = min(r0 + 1, p.h - 1), c1 = min(c0 + 1, p.w - 1);
            const float ly = sy - (float)r0, lx = sx - (float)c0;
            ro[d][0] = (r0 - r_lo) * p.w * 18;
            ro[d][1] = (r1 - r_lo) * p.w * 18;
            co[d][0] = c0 * 18 + half;
            co[d][1] = c1 * 18 + half;
            wy[d][0] = vy ? 1.f - ly : 0.f;
            wy[d][1] = vy ? ly : 0.f;
            wx[d][0] = vx ? 1.f - lx : 0.f;
            wx[d][1] = vx ? lx : 0.f;
        }
        float4 acc = p.bias ? *reinterpret_cast<const float4 *>(p.bias + n0 + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int t2 = 2 * (dy * 3 + dx);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const float wgt = wy[dy][a] * wx[dx][c];
                        const float4 v = src4[ro[dy][a] + co[dx][c] + t2];
                        acc.x = fmaf(wgt, v.x, acc.x);
                        acc.y = fmaf(wgt, v.y, acc.y);
                        acc.z = fmaf(wgt, v.z, acc.z);
                        acc.w = fmaf(wgt, v.w, acc.w);
                    }
            }
        if (p.relu) {
            acc.x = fmaxf(acc.x, 0.f);
            acc.y = fmaxf(acc.y, 0.f);
            acc.z = fmaxf(acc.z, 0.f);
            acc.w = fmaxf(acc.w, 0.f);
        }
        float *o = OUT_C8 ? p.y + ((((size_t)b * (p.N / 8) + blk) * p.H + y) * p.W + x) * 8 + 4 * half
                          : p.y + (((size_t)b * p.H + y) * p.W + x) * p.N + n0 + 4 * half;
        *reinterpret_cast<float4 *>(o) = acc;
    }
}

// output rows per band: the largest divisor-free choice whose source rows fit the LDS budget
int gather_band_rows(int w, int H, float hscale, int *rows_bound)
{
    int best = 1, bound = 3;
    for (int rb = 1; rb <= H; ++rb) {
        const int rows = (int)floorf((float)(rb + 1) * hscale) + 3;
        if ((size_t)rows * w * 288 > kGatherLdsBytes) break;
        best = rb;
        bound = rows;
    }
    // even bands: the last one is not a sliver
    const int bands = mpsr::ceil_div(H, best);
    best = mpsr::ceil_div(H, bands);
    bound = (int)floorf((float)(best + 1) * hscale) + 3;
    *rows_bound = bound;
    return best;
}

}  // namespace

namespace mpsr {

size_t upconv_weight_floats(int C, int N) { return (size_t)9 * N * C; }
size_t upconv_z_floats(long long Msrc, int N) { return (size_t)Msrc * 9 * (size_t)N; }

// x (B,h,w,C) NHWC, 3x3 filter (N, 9 C), output (B,OH,OW,N): N a multiple of 128 (whole GEMM parts), the source map a
// shape the pointwise kernel takes, 32-bit byte offsets inside z parts
bool upconv_applies(int B, int h, int w, int C, int OH, int OW, int N)
{
    const long long M = (long long)B * h * w;
    return B > 0 && h >= 1 && w >= 1 && OH >= 1 && OW >= 1 && N >= 128 && N % 128 == 0 && C % 4 == 0 &&
           pointwise_applies(M, C, PART) && (size_t)3 * w * 288 <= kGatherLdsBytes && B <= 65535 && N / 8 <= 65535 &&
           (long long)B * OH * OW * N * 4 < 0x7fffffffffLL;
}

int conv3x3_upsampled(const float *x, int B, int h, int w, int C, int OH, int OW, int align_corners, const float *g,
                      const float *bias, int relu, float *y, int N, int out_c8, float *z, size_t z_floats, float *ws,
                      size_t ws_floats, hipStream_t s)
{
    MPSR_REQUIRE(upconv_applies(B, h, w, C, OH, OW, N), "conv3x3_upsampled: unsupported shape (B=%d %dx%dx%d -> %dx%dx%d)",
                 B, h, w, C, OH, OW, N);
    const long long M = (long long)B * h * w;
    if (!z || z_floats < upconv_z_floats(M, N))
        return fail(MPSR_ERR_WORKSPACE, "conv3x3_upsampled: z scratch holds %zu floats, needs %zu", z_floats,
                    upconv_z_floats(M, N));
    // re-ordered filter rows: the caller's filter cache (mpsr_net_opts) if the network entry point offered a slot
    float *wp = ws;
    bool ready = false;
    if (g_filter_cache_slot.w == g && g_filter_cache_slot.u && g_filter_cache_slot.floats >= upconv_weight_floats(C, N)) {
        wp = g_filter_cache_slot.u;
        ready = g_filter_cache_slot.ready;
    } else if (!ws || ws_floats < upconv_weight_floats(C, N)) {
        g_filter_cache_slot = FilterCacheSlot();
        return fail(MPSR_ERR_WORKSPACE, "conv3x3_upsampled: scratch holds %zu floats, needs %zu", ws_floats,
                    upconv_weight_floats(C, N));
    }
    g_filter_cache_slot = FilterCacheSlot();
    if (!ready) {
        const long long total = (long long)9 * N * (C / 4);
        hipLaunchKernelGGL(upconv_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, N, C, wp);
        MPSR_CHECK_LAUNCH("upconv_weights_kernel");
    }
    const int parts = N / 128;
    for (int pidx = 0; pidx < parts; ++pidx) {
        const int rc = conv1x1_pointwise(x, M, C, wp + (size_t)pidx * PART * C, nullptr, nullptr, 0,
                                         z + (size_t)pidx * M * PART, PART, s);
        if (rc) return rc;
    }
    UpcParams p;
    p.z = z; p.bias = bias; p.y = y;
    p.part_stride = (size_t)M * PART;
    p.h = h; p.w = w; p.H = OH; p.W = OW; p.N = N; p.relu = relu;
    p.hscale = (align_corners && OH > 1) ? (float)(h - 1) / (float)(OH - 1) : (float)h / (float)OH;
    p.wscale = (align_corners && OW > 1) ? (float)(w - 1) / (float)(OW - 1) : (float)w / (float)OW;
    int rows_bound = 0;
    p.RB = gather_band_rows(w, OH, p.hscale, &rows_bound);
    const size_t lds = (size_t)rows_bound * w * 288;
    const dim3 grid((unsigned)(N / 8), (unsigned)ceil_div(OH, p.RB), (unsigned)B);
    if (out_c8) {
        MPSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(upconv_gather_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(upconv_gather_kernel<true>, grid, dim3(256), lds, s, p);
    } else {
        MPSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(upconv_gather_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(upconv_gather_kernel<false>, grid, dim3(256), lds, s, p);
    }
    MPSR_CHECK_LAUNCH("upconv_gather_kernel");
    return MPSR_OK;
}

}  // namespace mpsr

extern "C" size_t mpsr_conv3x3_upsampled_scratch_floats(int B, int h, int w, int C, int N)
{
    if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || N <= 0) return 0;
    return mpsr::align_up(mpsr::upconv_z_floats((long long)B * h * w, N), 64) + mpsr::upconv_weight_floats(C, N);
}

extern "C" int mpsr_conv3x3_upsampled_f32(const float *x, int B, int h, int w, int C, int OH, int OW, int align_corners,
                                          const float *weights, const float *bias, int relu, float *y, int N,
                                          float *ws, size_t ws_floats, mpsr_stream_t stream)
{
    MPSR_REQUIRE(B >= 0 && h > 0 && w > 0 && C > 0 && OH > 0 && OW > 0 && N > 0, "conv3x3_upsampled: bad shape");
    if (B == 0) return MPSR_OK;
    MPSR_REQUIRE(x && weights && y && ws, "conv3x3_upsampled: null pointer");
    if (!mpsr::upconv_applies(B, h, w, C, OH, OW, N))
        return mpsr::fail(MPSR_ERR_UNSUPPORTED,
                          "conv3x3_upsampled: needs N %% 128 == 0, C %% 64 == 0 and >= 128 (B=%d %dx%dx%d -> %dx%dx%d); use "
                          "mpsr_resize_bilinear + mpsr_conv2d_nhwc_f32",
                          B, h, w, C, OH, OW, N);
    const size_t zf = mpsr::align_up(mpsr::upconv_z_floats((long long)B * h * w, N), 64);
    if (ws_floats < zf + mpsr::upconv_weight_floats(C, N))
        return mpsr::fail(MPSR_ERR_WORKSPACE, "conv3x3_upsampled: scratch holds %zu floats, needs %zu", ws_floats,
                          zf + mpsr::upconv_weight_floats(C, N));
    mpsr::g_filter_cache_slot = mpsr::FilterCacheSlot();
    return mpsr::conv3x3_upsampled(x, B, h, w, C, OH, OW, align_corners, weights, bias, relu, y, N, 0, ws, zf, ws + zf,
                                   ws_floats - zf, mpsr::as_stream(stream));
}
