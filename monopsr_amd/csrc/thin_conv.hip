// 3x3 convolutions with a THIN side: the backward pass of the xyz-map head (reference graph net_builder.py:88-92, a 3x3
// convolution from the decoder's 128-channel feature map to 3 output channels -- padded to 4 here -- on the 48x48 map;
// its gradients are part of every training step: core/trainer.py:76-81).
//   * weight gradient  dW[n][t][c] = sum_p dy[p][n] x[p + off_t][c]   (x: B x 48 x 48 x 128 = 302 MB, dy: 9 MB)
//   * data gradient    dx[p][c]    = sum_t sum_n dy[p + off_t][n] wd[c][t][n]   (the forward form with C = 4 inputs)
// Both are ~5 GFLOP over a 302 MB tensor: memory-bound by two orders of magnitude.  On the general kernels they were
// the two worst launches of the training step: the implicit-GEMM weight gradient reduces 589 824 pixels into a
// 4 x 1152 output through 32-wide MFMA tiles (1.32 ms), the data gradient ran 128 x 128 tiles with K = 36 (0.37 ms).
// Here one WAVE owns 64 channels of the wide tensor (lane = channel) and walks a strip of six image rows pixel by
// pixel; the thin tensor's rows of the strip (+ one halo row above and below, + one halo column left and right,
// zero-filled) sit in an LDS window of the wave, and a pixel's 3 x 3 x 4 thin values reach every lane as nine
// broadcast ds_read_b128 -- three per pixel, the window slides in registers.  36 multiply-adds per pixel and lane,
// 256 contiguous bytes of the wide tensor per wave and pixel, six pixels requested ahead.
//   * weight gradient: the 36 sums (9 taps x 4 outputs) stay in registers for the wave's whole life; the waves of a
//     workgroup that own the same 64 channels are added through LDS and leave as one atomic per sum and workgroup
//     (<= 512 workgroups: same-address atomics serialise -- batchnorm.hip).  db rides along (sum of the staged rows).
//   * data gradient / thin-input forward: the 36 weights of the lane's output channel live in registers; one store
//     per pixel and lane.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <algorithm>
#include <atomic>

#include "common.h"

namespace {

constexpr int TCOLS = 64;  // staged columns: W + 2 <= 64 (lane = staged column)
constexpr int TROWS = 6;   // image rows per strip
constexpr int TWAVES = 8;  // waves per workgroup
constexpr int TAHEAD = 6;  // pixels of the wide tensor requested ahead

struct ThinParams {
    const float *wide;  // (B, H, W, C): x of the weight gradient
    const float *thin;  // (B, H, W, 4)
    const float *w;     // thin-input forward: (N, 9, 4) filters
    const float *bias;
    float *out;  // weight gradient: dw (4, 9, C), += ; forward: (B, H, W, N)
    float *db;   // weight gradient: (4), += ; may be null
    int B, H, W, C;  // C: channels of the wide side (the forward's N)
    int strips;      // ceil(H / TROWS)
    int relu;
};

// the wave's window: rows y0 - 1 .. y0 + TROWS of image b, columns -1 .. W (zero outside the image)
__device__ __forceinline__ float4 stage_thin(float4 (*win)[TCOLS], const float *thin, int b, int y0, int H, int W, int lane)
{
    float4 inside = make_float4(0.f, 0.f, 0.f, 0.f);  // sum of this lane's column over the strip's own rows
    const int x = lane - 1;
#pragma unroll
    for (int r = 0; r < TROWS + 2; ++r) {
        const int y = y0 - 1 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < H && x >= 0 && x < W) v = reinterpret_cast<const float4 *>(thin)[((size_t)b * H + y) * W + x];
        win[r][lane] = v;
        if (r >= 1 && r <= TROWS) {
            inside.x += v.x; inside.y += v.y; inside.z += v.z; inside.w += v.w;
        }
    }
    return inside;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// dW[n][t][c] += sum_p dy[p][n] x[p + off_t][c]; wave w of a workgroup owns channel group w % ncg for its whole life.
__global__ __launch_bounds__(TWAVES * 64) void thin_wgrad_kernel(const ThinParams p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ncg = p.C >> 6;
    const int cg = wave % ncg, sub = wave / ncg, nsub = TWAVES / ncg;
    float4(*win)[TCOLS] = reinterpret_cast<float4(*)[TCOLS]>(lds + (size_t)wave * (TROWS + 2) * TCOLS * sizeof(float4));
    float acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[t][n] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const int units = p.B * p.strips;
    for (int u = blockIdx.x * nsub + sub; u < units; u += gridDim.x * nsub) {
        const int b = u / p.strips, y0 = (u % p.strips) * TROWS;
        const int rows = min(TROWS, p.H - y0);
        const float4 ins = stage_thin(win, p.thin, b, y0, p.H, p.W, lane);
        if (cg == 0) {  // (rows past the image were staged as zeros)
            bsum.x += ins.x; bsum.y += ins.y; bsum.z += ins.z; bsum.w += ins.w;
        }
        __builtin_amdgcn_wave_barrier();
        for (int yy = 0; yy < rows; ++yy) {
            const float *xrow = p.wide + (((size_t)b * p.H + y0 + yy) * p.W) * p.C + cg * 64 + lane;
            float xn[TAHEAD];
#pragma unroll
            for (int k = 0; k < TAHEAD; ++k) xn[k] = k < p.W ? xrow[(size_t)k * p.C] : 0.f;
            // window columns xx, xx + 1, xx + 2 (staged coordinates) of rows yy .. yy + 2 in three rotating register
            // slots: pixel k of a block of six keeps column xx + j in slot (k + j) % 3 and fetches column xx + 2
            static_assert(TAHEAD % 3 == 0, "the slot rotation must close over a block");
            float4 col[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                col[0][r] = win[yy + r][0];
                col[1][r] = win[yy + r][1];
            }
            for (int x0 = 0; x0 < p.W; x0 += TAHEAD) {
                float xv[TAHEAD];
#pragma unroll
                for (int k = 0; k < TAHEAD; ++k) {
                    xv[k] = xn[k];
                    const int xa = x0 + TAHEAD + k;
                    xn[k] = xa < p.W ? xrow[(size_t)xa * p.C] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < TAHEAD; ++k) {
                    const int xx = x0 + k;
                    if (xx < p.W) {  // (uniform)
#pragma unroll
                        for (int r = 0; r < 3; ++r) col[(k + 2) % 3][r] = win[yy + r][xx + 2];
                        // tap (ty, tx): output pixel (y - ty + 1, xx - tx + 1) = staged (yy - ty + 2, xx - tx + 2)
#pragma unroll
                        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                            for (int tx = 0; tx < 3; ++tx) {
                                const float4 d = col[(k + 2 - tx) % 3][2 - ty];
                                float *a = acc[ty * 3 + tx];
                                a[0] = fmaf(d.x, xv[k], a[0]);
                                a[1] = fmaf(d.y, xv[k], a[1]);
                                a[2] = fmaf(d.z, xv[k], a[2]);
                                a[3] = fmaf(d.w, xv[k], a[3]);
                            }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- the workgroup's waves of one channel group summed through LDS, one atomic per sum
    __syncthreads();
    float *part = reinterpret_cast<float *>(lds);  // [wave][t * 4 + n][lane]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int n = 0; n < 4; ++n) part[(wave * 36 + t * 4 + n) * 64 + lane] = acc[t][n];
    __syncthreads();
    const int outputs = ncg * 36 * 64;
    for (int o = threadIdx.x; o < outputs; o += TWAVES * 64) {
        const int l = o & 63, q = (o >> 6) % 36, g = o / (36 * 64);
        float s = 0.f;
        for (int k = 0; k < nsub; ++k) s += part[((k * ncg + g) * 36 + q) * 64 + l];
        const int t = q >> 2, n = q & 3;
        unsafeAtomicAdd(&p.out[((size_t)n * 9 + t) * p.C + g * 64 + l], s);
    }
    if (p.db && cg == 0) {  // (every wave of channel group 0 staged different strips)
        const float s0 = wave_sum(bsum.x), s1 = wave_sum(bsum.y), s2 = wave_sum(bsum.z), s3 = wave_sum(bsum.w);
        if (lane < 4) unsafeAtomicAdd(&p.db[lane], lane == 0 ? s0 : lane == 1 ? s1 : lane == 2 ? s2 : s3);
    }
}

// out[p][c] = act(bias[c] + sum_t sum_n thin[p + off_t][n] w[c][t][n]); unit = (image, strip, channel group)
__global__ __launch_bounds__(TWAVES * 64) void thin_input_conv_kernel(const ThinParams p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ncg = p.C >> 6;
    float4(*win)[TCOLS] = reinterpret_cast<float4(*)[TCOLS]>(lds + (size_t)wave * (TROWS + 2) * TCOLS * sizeof(float4));
    const int units = p.B * p.strips * ncg;
    const float lo = p.relu ? 0.f : -__builtin_inff();
    for (int u = blockIdx.x * TWAVES + wave; u < units; u += gridDim.x * TWAVES) {
        const int cg = u % ncg, s = u / ncg;
        const int b = s / p.strips, y0 = (s % p.strips) * TROWS;
        const int rows = min(TROWS, p.H - y0);
        const int c = cg * 64 + lane;
        float4 wt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = reinterpret_cast<const float4 *>(p.w)[(size_t)c * 9 + t];
        const float bias = p.bias ? p.bias[c] : 0.f;
        stage_thin(win, p.thin, b, y0, p.H, p.W, lane);
        __builtin_amdgcn_wave_barrier();
        for (int yy = 0; yy < rows; ++yy) {
            float *orow = p.out + (((size_t)b * p.H + y0 + yy) * p.W) * p.C + c;
            float4 col[3][3];  // (rotating slots as in the weight gradient)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                col[0][r] = win[yy + r][0];
                col[1][r] = win[yy + r][1];
            }
            for (int x0 = 0; x0 < p.W; x0 += 3) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int xx = x0 + k;
                    if (xx < p.W) {  // (uniform)
#pragma unroll
                        for (int r = 0; r < 3; ++r) col[(k + 2) % 3][r] = win[yy + r][xx + 2];
                        // tap (ty, tx) reads input pixel (y + ty - 1, xx + tx - 1) = staged (yy + ty, xx + tx)
                        float a = bias;
#pragma unroll
                        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                            for (int tx = 0; tx < 3; ++tx) {
                                const float4 d = col[(k + tx) % 3][ty];
                                const float4 g = wt[ty * 3 + tx];
                                a = fmaf(d.x, g.x, a);
                                a = fmaf(d.y, g.y, a);
                                a = fmaf(d.z, g.z, a);
                                a = fmaf(d.w, g.w, a);
                            }
                        orow[(size_t)xx * p.C] = fmaxf(a, lo);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

constexpr size_t kWinBytes = (size_t)TWAVES * (TROWS + 2) * TCOLS * sizeof(float4);  // 65536
constexpr size_t kPartBytes = (size_t)TWAVES * 36 * 64 * sizeof(float);              // 73728

}  // namespace

namespace mpsr {

static std::atomic<int> g_thin{1};  // mpsr_debug_set_thin_conv: 0 = these layers stay on the general kernels (A/B)

bool thin_wgrad_applies(int B, int H, int W, int C, int N, int KH, int KW, int dilation)
{
    return g_thin.load() && KH == 3 && KW == 3 && dilation == 1 && N == 4 && (C == 64 || C == 128 || C == 256) &&
           W + 2 <= TCOLS && B > 0 && H > 0 && (long long)B * H * W * C * 4 < 0x7fffffffffLL;
}

int thin_wgrad(const float *x, const float *dy, int B, int H, int W, int C, float *dw, float *db, hipStream_t s)
{
    ThinParams p{};
    p.wide = x; p.thin = dy; p.out = dw; p.db = db;
    p.B = B; p.H = H; p.W = W; p.C = C;
    p.strips = (H + TROWS - 1) / TROWS;
    const int nsub = TWAVES / (C >> 6);
    const long long units = (long long)B * p.strips;
    const int grid = (int)std::min<long long>(512, (units + nsub - 1) / nsub);
    const size_t ldsb = std::max(kWinBytes, kPartBytes);
    MPSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(thin_wgrad_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipLaunchKernelGGL(thin_wgrad_kernel, dim3(grid), dim3(TWAVES * 64), ldsb, s, p);
    MPSR_CHECK_LAUNCH("thin_wgrad_kernel");
    return MPSR_OK;
}

bool thin_input_conv_applies(int B, int H, int W, int C, int N, int KH, int KW, int dilation)
{
    return g_thin.load() && KH == 3 && KW == 3 && dilation == 1 && C == 4 && (N == 64 || N == 128 || N == 256) &&
           W + 2 <= TCOLS && B > 0 && H > 0;
}

int thin_input_conv(const float *x, int B, int H, int W, const float *w, const float *bias, int relu, float *y, int N,
                    hipStream_t s)
{
    ThinParams p{};
    p.thin = x; p.w = w; p.bias = bias; p.out = y;
    p.B = B; p.H = H; p.W = W; p.C = N; p.relu = relu;
    p.strips = (H + TROWS - 1) / TROWS;
    const long long units = (long long)B * p.strips * (N >> 6);
    const int grid = (int)std::min<long long>(2048, (units + TWAVES - 1) / TWAVES);
    MPSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(thin_input_conv_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWinBytes));
    hipLaunchKernelGGL(thin_input_conv_kernel, dim3(grid), dim3(TWAVES * 64), kWinBytes, s, p);
    MPSR_CHECK_LAUNCH("thin_input_conv_kernel");
    return MPSR_OK;
}

}  // namespace mpsr

extern "C" void mpsr_debug_set_thin_conv(int on) { mpsr::g_thin = on; }
