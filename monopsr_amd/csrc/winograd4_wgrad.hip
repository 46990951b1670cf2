// Weight gradient of a 3x3 stride-1 SAME convolution in the Winograd F(4x4,3x3) domain: the map decoder's four dense
// 3x3 layers in training (reference: TensorFlow autodiff of monopsr/builders/net_builder.py:70-89,
// monopsr/core/trainer.py:71-81).
//
//   forward    Y (4x4) = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A
//   gradient   dg = G^T [ sum over tiles and images of (A dY A^T) (.) (B^T d B) ] G
// i.e. per element position p (36 of them) one GEMM over the tiles t
//   dU_p[n][c] = sum_t Yh_p[t][n] * V_p[t][c],          Yh = A dY A^T (4x4 -> 6x6),  V = B^T d B (the forward's transform)
// with 36 products per (channel pair, tile of 16 pixels) where the direct weight gradient (backward.hip) has 144.
// Both operands are transformed on the fly, neither reaches HBM; dU (36 x N x C) is combined over tile slices with
// fp32 atomics and folded back by G^T . G in a second, tiny kernel.
//
// Workgroup = 512 threads = 8 waves, a 32 (n) x 32 (c) block of all 36 positions, a slice of the tiles; K step = 8
// tiles.  Waves 0-3 produce Yh: thread = (tile of the step, n): 16 loads of its 4x4 dY block, 80 operations, 36
// ds_write_b32; waves 4-7 produce V: thread = (tile, c): 36 loads of the 6x6 patch (out-of-image pixels by an
// out-of-range offset), 144 operations, 36 stores -- every SIMD hosts one wave of each kind.  Both land in LDS as
// [position][8 tiles][32 rows n or c], double buffered (2 x 74 KB); a wave then multiplies its 5 (waves 0-3) or 4
// (waves 4-7) positions: four ds_read2_b32 per 4 MFMAs.  A thread transforms the
// patch it requested during the previous step, stores it for the next one, multiplies, and requests the patch after
// next into the same registers: one barrier per step.
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace w4g {
constexpr int KT = 8;                 // tiles per K step
constexpr int POSF = 32 * KT;         // floats per position of one operand (256)
constexpr int OPF = 36 * POSF;        // one operand of a stage (9216 floats = 36 KB)
constexpr int STAGEF = 2 * OPF;       // Yh then V
constexpr int LDSF = 2 * STAGEF;      // 36864 floats = 144 KB
constexpr unsigned OOB = 0x80000000u;
}  // namespace w4g

struct W4gParams {
    const float *x, *dy;
    float *du;
    int B, H, W, C, N, th, tw, T;
    int nblocks, cblocks, nslices, steps;  // steps: K steps (of 8 tiles) per slice
    mpsr::FastDiv fd_tpi, fd_tw;
    unsigned xbytes, dybytes;
};

// B^T (6x6) applied to a 6-vector, in place (winograd4.hip's): 12 operations
__device__ __forceinline__ void g_bt6(float &d0, float &d1, float &d2, float &d3, float &d4, float &d5)
{
    const float a = fmaf(-4.f, d2, d4), b = fmaf(-4.f, d1, d3), c = d4 - d2, e = d3 - d1;
    const float t0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
    const float t5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
    d0 = t0;
    d1 = a + b;
    d2 = a - b;
    d3 = fmaf(2.f, e, c);
    d4 = fmaf(-2.f, e, c);
    d5 = t5;
}
// A (6x4, the transpose of the forward's A^T) applied to a 4-vector: 8 operations
__device__ __forceinline__ void g_a6(float d0, float d1, float d2, float d3, float &o0, float &o1, float &o2, float &o3,
                                     float &o4, float &o5)
{
    const float e = d0 + d2, f = d1 + d3, g = fmaf(4.f, d2, d0), t = fmaf(4.f, d3, d1);
    o0 = d0;
    o1 = e + f;
    o2 = e - f;
    o3 = fmaf(2.f, t, g);
    o4 = fmaf(-2.f, t, g);
    o5 = d3;
}

// The body of a wave, one straight-line copy per role (VPROD: the wave transforms x patches, otherwise dY blocks) and
// number of positions it multiplies (NPOS), selected once in the kernel: no role test inside the K loop.
template <bool VPROD, int NPOS>
__device__ __forceinline__ void w4g_body(const W4gParams &p, float *lds, int n0, int c0, int t_begin, int pos0)
{
    using namespace w4g;
    const int tid = threadIdx.x, lane = tid & 63;
    // tile of the step and row (n or c) of the operand: a wave = 2 tiles x all 32 rows, rows fastest -- every load is two
    // whole 128-byte lines (8 rows x 8 tiles per wave made four waves fetch each line: the L2 -> L1 path, not the
    // matrix pipe, then set the step time), and with the operands kept as [position][tile][row] in LDS its 64 stores
    // of one position are 64 consecutive floats
    const int lt = (tid & 255) >> 5, pc = tid & 31;
    const int tpi = p.th * p.tw;
    const unsigned shift = (unsigned)(p.W + 1) * (unsigned)p.C * 4u;  // x descriptor moved back by one row + one pixel
    char *xback = const_cast<char *>(reinterpret_cast<const char *>(p.x)) - shift;
    constexpr int NPA = VPROD ? 36 : 16;
    float pa[NPA];
    // requests the patch of K step `step` (all-zero past the slice's last step or the last tile)
    auto request = [&](int step) __attribute__((always_inline)) {
        const bool live = step < p.steps;
        const int t = t_begin + (live ? step : 0) * KT + lt;
        const int img = mpsr::fdiv(t, p.fd_tpi), rem = t - img * tpi;
        const int ty = mpsr::fdiv(rem, p.fd_tw), tx = rem - ty * p.tw;
        const bool in = t < p.T;
        if constexpr (VPROD) {
            const __amdgpu_buffer_rsrc_t rr =
                __builtin_amdgcn_make_buffer_rsrc(xback, 0, live ? (int)(p.xbytes + shift) : 0, 0x00020000);
            const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
            // (relative to the moved-back descriptor this is the offset of pixel (y0, x0))
            const unsigned abase = (unsigned)(((img * p.H + y0 + 1) * p.W + x0 + 1) * p.C + c0 + pc) * 4u;
            // only the ring of the patch can leave the image: nine offsets (this thread's pixel (0, 0), or out of range)
            const bool rowc[3] = {in && y0 >= 0, in, in && y0 + 5 < p.H};
            const bool colc[3] = {x0 >= 0, true, x0 + 5 < p.W};
            unsigned vo[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) vo[a][b] = (rowc[a] && colc[b]) ? abase : OOB;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int s = 0; s < 6; ++s)
                    pa[6 * r + s] = __builtin_bit_cast(
                        float, __builtin_amdgcn_raw_buffer_load_b32(rr, vo[r == 0 ? 0 : r == 5 ? 2 : 1][s == 0 ? 0 : s == 5 ? 2 : 1],
                                                                    (r * p.W + s) * p.C * 4, 0));
        } else {
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(p.dy), 0, live ? (int)p.dybytes : 0, 0x00020000);
            const unsigned abase = in ? (unsigned)(((img * p.H + 4 * ty) * p.W + 4 * tx) * p.N + n0 + pc) * 4u : OOB;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    pa[4 * r + s] = __builtin_bit_cast(
                        float, __builtin_amdgcn_raw_buffer_load_b32(rr, abase, (r * p.W + s) * p.N * 4, 0));
        }
    };
    // [stage][operand][position][8 tiles][32 rows]
    float *wr = lds + (VPROD ? OPF : 0) + lt * 32 + pc;
    auto transform_store = [&](int stage) __attribute__((always_inline)) {
        float *w0 = wr + stage * STAGEF;
        if constexpr (VPROD) {
#pragma unroll
            for (int s = 0; s < 6; ++s) g_bt6(pa[s], pa[6 + s], pa[12 + s], pa[18 + s], pa[24 + s], pa[30 + s]);
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                g_bt6(pa[6 * r], pa[6 * r + 1], pa[6 * r + 2], pa[6 * r + 3], pa[6 * r + 4], pa[6 * r + 5]);
#pragma unroll
                for (int j = 0; j < 6; ++j) w0[(6 * r + j) * POSF] = pa[6 * r + j];
            }
        } else {
            float m[6][4];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                g_a6(pa[s], pa[4 + s], pa[8 + s], pa[12 + s], m[0][s], m[1][s], m[2][s], m[3][s], m[4][s], m[5][s]);
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                float o[6];
                g_a6(m[r][0], m[r][1], m[r][2], m[r][3], o[0], o[1], o[2], o[3], o[4], o[5]);
#pragma unroll
                for (int j = 0; j < 6; ++j) w0[(6 * r + j) * POSF] = o[j];
            }
        }
    };

    // fragment of a lane = (row lane & 31, tiles 4 h .. 4 h + 3 of the step, h = lane >> 5): four floats 32 apart
    const float *rd = lds + pos0 * POSF + (lane >> 5) * 128 + (lane & 31);
    f32x16 acc[NPOS];
#pragma unroll
    for (int q = 0; q < NPOS; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    auto multiply = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NPOS; ++q) {
            const float *ra = rd + stage * STAGEF + q * POSF, *rb = ra + OPF;
            const float4 a = make_float4(ra[0], ra[32], ra[64], ra[96]);
            const float4 b = make_float4(rb[0], rb[32], rb[64], rb[96]);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[q], 0, 0, 0);
        }
    };

    request(0);
    transform_store(0);
    request(1);
    __syncthreads();
    // A step = the products of the current stage + the transform of the next stage's patch (independent of each
    // other) + the request of the patch after next into the registers the transform just freed.  The two waves of a
    // SIMD take the first two in opposite order: while the x-side wave transforms (vector ALU, LDS stores) the dY-side
    // wave multiplies, then they swap.  (Left to hipcc's scheduler, which weaves the transform's arithmetic between
    // the MFMAs: pinning the phases with sched_barrier measured 8-12 % slower; a second patch register set -- requests
    // two steps ahead -- changed nothing.)
    for (int s = 0; s < p.steps; s += 2) {
        if constexpr (VPROD) {
            transform_store(1);  // the patch of step s + 1 (zeros past the end)
            multiply(0);
        } else {
            multiply(0);
            transform_store(1);
        }
        request(s + 2);
        __syncthreads();
        if constexpr (VPROD) {
            transform_store(0);
            multiply(1);
        } else {
            multiply(1);
            transform_store(0);
        }
        request(s + 3);
        __syncthreads();
    }
    // (the 16-pass MFMA needs 18 wait states before its result is read; explicit as in conv_mfma.hip)
#pragma unroll
    for (int q = 0; q < NPOS; ++q) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[q]));
    // accumulator element e of a lane: row (n) = (e & 3) + 8 (e >> 2) + 4 (lane >> 5), column (c) = lane & 31
#pragma unroll
    for (int q = 0; q < NPOS; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = n0 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            unsafeAtomicAdd(&p.du[((size_t)(pos0 + q) * p.N + n) * p.C + c0 + (lane & 31)], acc[q][e]);
        }
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino4_wgrad_kernel(const W4gParams p)
{
    using namespace w4g;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // XCD x (workgroup b runs on XCD b % 8: speed only) takes the tile slices x, x + 8, ...; the (n, c) blocks of one
    // slice are consecutive workgroups of it, so the slice's pixels come from HBM once
    const int xcd = blockIdx.x & 7, l_ = blockIdx.x >> 3;
    const int nblk = p.nblocks * p.cblocks;
    const int blk = l_ % nblk, slice = (l_ / nblk) * 8 + xcd;
    if (slice >= p.nslices) return;  // block-uniform
    const int n0 = (blk / p.cblocks) * 32, c0 = (blk % p.cblocks) * 32;
    const int t_begin = slice * p.steps * KT;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // waves 0-3 (dY side, the lighter transform) multiply five positions each (0..19), waves 4-7 four (20..35)
    if (wave < 4) w4g_body<false, 5>(p, lds, n0, c0, t_begin, 5 * wave);
    else w4g_body<true, 4>(p, lds, n0, c0, t_begin, 20 + 4 * (wave - 4));
}

// dw[n][(a * 3 + b) * C + c] += (G^T dU G)[a][b]; one thread per (n, c), float64 inside
__global__ __launch_bounds__(256) void wino4_wgrad_finish_kernel(const float *__restrict__ du, int N, int C,
                                                                 float *__restrict__ dw)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * C) return;
    const int n = (int)(i / C), c = (int)(i - (long long)n * C);
    auto gt = [](const double *m, double *o) {  // G^T (3x6) applied to a 6-vector
        o[0] = m[0] / 4.0 - (m[1] + m[2]) / 6.0 + (m[3] + m[4]) / 24.0;
        o[1] = (m[2] - m[1]) / 6.0 + (m[3] - m[4]) / 12.0;
        o[2] = -(m[1] + m[2]) / 6.0 + (m[3] + m[4]) / 6.0 + m[5];
    };
    double t[3][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double col[6], o[3];
#pragma unroll
        for (int r = 0; r < 6; ++r) col[r] = (double)du[((size_t)(6 * r + j) * N + n) * C + c];
        gt(col, o);
#pragma unroll
        for (int a = 0; a < 3; ++a) t[a][j] = o[a];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double o[3];
        gt(t[a], o);
#pragma unroll
        for (int b = 0; b < 3; ++b) dw[(size_t)n * 9 * C + (size_t)(a * 3 + b) * C + c] += (float)o[b];
    }
}

}  // namespace

namespace mpsr {

// Shapes the Winograd weight gradient takes: 3x3 dense, the map divides into 4x4 blocks, 32-channel blocks, wide and
// large enough to pay, 32-bit byte offsets.
bool winograd4_wgrad_applies(int B, int H, int W, int C, int N, int KH, int KW, int dilation)
{
    const long long M = (long long)B * H * W;
    // (measured: 14-22 % faster than the direct kernel on the decoder's layers, slower on a 96 -> 64 layer of 86 k pixels)
    return KH == 3 && KW == 3 && dilation == 1 && H % 4 == 0 && W % 4 == 0 && C % 32 == 0 && N % 32 == 0 && C >= 128 &&
           N >= 128 && M >= 131072 && M * C * 4 < 0x7f000000LL && M * N * 4 < 0x7f000000LL;
}
size_t winograd4_wgrad_scratch_floats(int C, int N) { return (size_t)36 * N * C; }

int conv3x3_wgrad_winograd4(const float *x, const float *dy, int B, int H, int W, int C, int N, float *dw, float *ws,
                            size_t ws_floats, hipStream_t s)
{
    using namespace w4g;
    MPSR_REQUIRE(winograd4_wgrad_applies(B, H, W, C, N, 3, 3, 1), "conv3x3_wgrad_winograd4: unsupported shape");
    const size_t need = winograd4_wgrad_scratch_floats(C, N);
    if (!ws || ws_floats < need)
        return fail(MPSR_ERR_WORKSPACE, "conv3x3_wgrad_winograd4: scratch holds %zu floats, needs %zu", ws_floats, need);
    W4gParams p;
    p.x = x; p.dy = dy; p.du = ws;
    p.B = B; p.H = H; p.W = W; p.C = C; p.N = N;
    p.th = H / 4; p.tw = W / 4;
    p.T = B * p.th * p.tw;
    p.nblocks = N / 32; p.cblocks = C / 32;
    p.fd_tpi = make_fastdiv(p.th * p.tw);
    p.fd_tw = make_fastdiv(p.tw);
    p.xbytes = (unsigned)((long long)B * H * W * C * 4);
    p.dybytes = (unsigned)((long long)B * H * W * N * 4);
    // one workgroup per CU (144 KB of LDS): about two rounds of slices, each an even number of 8-tile steps
    const int blocks = p.nblocks * p.cblocks;
    // (a multiple of 8: slice i runs on XCD i % 8, fewer than 8 slices would leave XCDs idle)
    int slices = (512 / blocks + 7) / 8 * 8;
    if (slices < 8) slices = 8;
    int steps = (int)(((long long)p.T + (long long)slices * KT - 1) / ((long long)slices * KT));
    steps = (steps + 1) / 2 * 2;
    p.steps = steps;
    p.nslices = (int)(((long long)p.T + (long long)steps * KT - 1) / ((long long)steps * KT));
    MPSR_CHECK_HIP(hipMemsetAsync(ws, 0, need * sizeof(float), s));
    MPSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_wgrad_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDSF * sizeof(float))));
    const unsigned grid = (unsigned)(((p.nslices + 7) / 8) * 8 * blocks);
    hipLaunchKernelGGL(wino4_wgrad_kernel, dim3(grid), dim3(512), LDSF * sizeof(float), s, p);
    MPSR_CHECK_LAUNCH("wino4_wgrad_kernel");
    const long long total = (long long)N * C;
    hipLaunchKernelGGL(wino4_wgrad_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ws, N, C, dw);
    MPSR_CHECK_LAUNCH("wino4_wgrad_finish_kernel");
    return MPSR_OK;
}

}  // namespace mpsr

namespace mpsr {  // winograd3_wgrad.hip: the atrous layers whose pixel sub-grids are single 3x3 tiles (block3's conv2)
bool winograd3_wgrad_applies(int B, int H, int W, int C, int N, int KH, int KW, int dilation);
int conv3x3_wgrad_winograd3(const float *x, const float *dy, int B, int H, int W, int C, int N, int dilation, float *dw,
                            float *db, hipStream_t s);
}  // namespace mpsr

static int g_wgrad_winograd = 1;  // mpsr_debug_set_wgrad_winograd
extern "C" void mpsr_debug_set_wgrad_winograd(int on) { g_wgrad_winograd = on; }

extern "C" size_t mpsr_conv2d_wgrad_scratch_floats(int B, int H, int W, int C, int N, int KH, int KW, int dilation)
{
    return mpsr::winograd4_wgrad_applies(B, H, W, C, N, KH, KW, dilation) ? mpsr::winograd4_wgrad_scratch_floats(C, N) : 0;
}

extern "C" int mpsr_conv2d_wgrad_ws_f32(const float *x, const float *dy, int B, int H, int W, int C, int N, int KH, int KW,
                                        int dilation, float *dw, float *db, float *ws, size_t ws_floats,
                                        mpsr_stream_t stream)
{
    if (g_wgrad_winograd && ws && B > 0 && x && dy && dw && mpsr::winograd4_wgrad_applies(B, H, W, C, N, KH, KW, dilation) &&
        ws_floats >= mpsr::winograd4_wgrad_scratch_floats(C, N)) {
        int rc = mpsr::conv3x3_wgrad_winograd4(x, dy, B, H, W, C, N, dw, ws, ws_floats, mpsr::as_stream(stream));
        if (rc) return rc;
        return db ? mpsr_bias_grad(dy, (long long)B * H * W, N, db, stream) : MPSR_OK;
    }
    // (the F(3x3,3x3) form of block3's atrous layers needs no scratch: it accumulates into dw itself)
    if (g_wgrad_winograd && B > 0 && x && dy && dw && mpsr::winograd3_wgrad_applies(B, H, W, C, N, KH, KW, dilation)) {
        // (the bias gradient rides along: position (1, 1) of A dY A^T is the tile's dY sum)
        return mpsr::conv3x3_wgrad_winograd3(x, dy, B, H, W, C, N, dilation, dw, db, mpsr::as_stream(stream));
    }
    return mpsr_conv2d_wgrad_f32(x, dy, B, H, W, C, N, KH, KW, dilation, dw, db, stream);
}
