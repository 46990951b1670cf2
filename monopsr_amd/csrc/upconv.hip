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
// (upconv_gather_kernel) is the 9-tap x 4-corner weighted sum above: one workgroup per (image, 8-channel block) rolls a
// window of z rows through LDS and every thread sums 36 float4 for an output pixel's 4 channels -- LDS / vector-ALU work
// on 1/9 of the GEMM's flops, reading z once and writing the layer's output once (channel-blocked for the F(4x4,3x3)
// layer that follows, or NHWC).
//
// Column order of z (ours to choose): j = (n / 8) * 72 + t * 8 + (n % 8) -- the nine taps of an 8-channel block are 288
// contiguous bytes per source pixel.  The GEMM's weight matrix W'[j][c] = g[n][t][c] is a re-ordering of the filter
// rows (upconv_weights_kernel; kept in the caller's filter cache when there is one).  9 N columns are cut into parts of
// 1152 = 9 column blocks of the pointwise kernel (its persistent grid then fills the chip: 504 of 512 workgroup slots),
// i.e. one part per 128 output channels.
#include <atomic>

#include "common.h"
#include "wino3_filter.h"

namespace mpsr {
bool pointwise_applies(long long M, int K, int N);
int conv1x1_pointwise(const float *x, long long M, int K, const float *w, const float *bias, const float *residual,
                      int relu, float *y, int N, hipStream_t s);
}  // namespace mpsr

namespace {

using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int PART = 1152;  // z columns per GEMM launch = 16 channel blocks x 9 taps x 8 channels
constexpr size_t kGatherLdsBytes = 80 * 1024;  // most LDS a gather workgroup takes (two per CU)

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
    int B, h, w, H, W, N, relu;
    int RB, cap;         // output rows per band; source rows the LDS ring holds (>= the rows one band reaches)
    mpsr::FastDiv capdiv;  // r % cap by multiply-high (common.h)
    float hscale, wscale;
};

// cache policy of the gather's z loads (bit 0) and result stores (bit 1): 1 = non-temporal (A/B builds)
#ifndef UPC_NT
#define UPC_NT 0
#endif
#ifndef UPC_DEFER
#define UPC_DEFER 0  // 1: a band's results wait in registers and leave after the next band's rows are committed (r06 A/B: slower)
#endif
constexpr int kPf = 8;  // float4 registers per thread that carry source pixels on their way into LDS
constexpr int kGrp = 4;  // output rows per thread and band whose results wait in registers until the band's stores
std::atomic<int> g_upconv_band{0};  // mpsr_debug_set_upconv_band: output rows per band of the rolling window (0 = 8)

// One workgroup per (8-channel block, image) walks the image in bands of RB output rows.  LDS is a ring of `cap` source
// rows, [row % cap][column][tap][8 channels]; while a band is summed the source rows the NEXT band adds are already on
// their way into registers, and move into the ring between two barriers -- every z row is read exactly once, and the
// loads overlap the arithmetic inside the workgroup.  Two things measured on the way (r04): an integer division costs
// hipcc ~35 vector instructions -- a version with index decodes per item / per load ran 1.5x longer than its arithmetic
// -- so nothing in the loops divides: thread = (row of a pass, output column x, half of the 8 channels) with blockDim.x =
// rpp * 2 W (a thread's column, its three tap columns, their LDS offsets and weights never change) and, as a loader,
// (pixel group, 16-byte piece of a pixel's 288 bytes) walking pixels by addition.  And a load -> LDS-store loop waits for
// every load in turn (12 round trips per workgroup were 3/4 of the first version's time): loads are issued kPf at a time
// into registers, then stored.
template <bool OUT_C8>
__global__ __launch_bounds__(512) void upconv_gather_kernel(const UpcParams p)
{
    extern __shared__ __attribute__((aligned(16))) float4 src4[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    // Workgroup i runs on XCD i % 8 (speed only): XCD x takes the images x, x + 8, ... and the N / 8 channel blocks of an
    // image back to back.  A block's 288-byte slice of a pixel straddles cache lines it shares with its neighbours, and
    // an image's z slab (0.7 - 2.7 MB) is read by all its blocks: one L2 then fetches every line once (with the blocks
    // of an image spread over the eight L2s the same bytes came in 1.44 times).
    const int nblk = p.N >> 3;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int bq = jx / nblk, blk = jx - bq * nblk, b = bq * 8 + xcd;
    if (b >= p.B) return;  // block-uniform
    const int lpr = 2 * p.W, rpp = nthr / lpr;
    const int rr = tid / lpr, l = tid - rr * lpr, x = l >> 1, half = l & 1;
    const int n0 = blk * 8;
    const int rowf4 = p.w * 18;  // float4 per source row
    auto slot = [&](int r) __attribute__((always_inline)) { return r - p.cap * mpsr::fdiv(r, p.capdiv); };  // r % cap
    const float *zp = p.z + (size_t)(blk >> 4) * p.part_stride + (size_t)(blk & 15) * 72 + (size_t)b * p.h * p.w * PART;
    // loader role: piece `lpiece` of the pixels lgrp, lgrp + lgroups, ... of a block of whole rows (row-major, rows are
    // contiguous in z); (lrow0, lcol0) = this thread's first pixel, (drow, dcol) = the step of lgroups pixels
    const int lgroups = nthr / 18, lgrp = tid / 18, lpiece = tid - lgrp * 18;
    const bool loader = lgrp < lgroups;
    const int lrow0 = lgrp / p.w, lcol0 = lgrp - lrow0 * p.w;
    const int drow = lgroups / p.w, dcol = lgroups - drow * p.w;
    f32x4 pf[kPf];
    // pixels [first, first + kPf * lgroups) of the block of rows that starts at row ra, npix pixels long
    auto issue = [&](int ra, int npix, int first) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < kPf; ++j) {
            // (an unconditional load through a selected address -- unused slots re-read one cached line; a conditional
            // load, or HIP's float4 struct as the element type, sends the register array to scratch memory)
            const int pix = first + lgrp + j * lgroups;
#ifdef UPC_SKIP_LOAD  // (timing experiments only -- tools/gather_bench.py: every request re-reads one cached line)
            const bool live = false;
#else
            const bool live = loader && pix < npix;
#endif
#if UPC_NT & 1
            pf[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(live ? zp + ((size_t)ra * p.w + pix) * PART + 4 * lpiece : zp));
#else
            pf[j] = *reinterpret_cast<const f32x4 *>(live ? zp + ((size_t)ra * p.w + pix) * PART + 4 * lpiece : zp);
#endif
        }
    };
    // (row, col) = position of this thread's pixel `first + lgrp` relative to row ra, kept by the caller
    auto commit = [&](int ra, int npix, int first, int &row, int &col) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < kPf; ++j) {
            const int pix = first + lgrp + j * lgroups;
            if (loader && pix < npix)
                reinterpret_cast<f32x4 *>(src4)[slot(ra + row) * rowf4 + col * 18 + lpiece] = pf[j];
            row += drow;
            col += dcol;
            if (col >= p.w) {
                col -= p.w;
                ++row;
            }
        }
    };

    // source rows band k's taps (output rows k RB - 1 .. (k + 1) RB, clamped) interpolate between
    auto band_rows = [&](int k, int &lo, int &hi) __attribute__((always_inline)) {
        const int y0 = k * p.RB, yl = min(y0 + p.RB, p.H) - 1;
        lo = (int)floorf((float)max(y0 - 1, 0) * p.hscale);
        hi = min((int)floorf((float)min(yl + 1, p.H - 1) * p.hscale) + 1, p.h - 1);
    };
    // tap column d - 1 of this thread's output column: LDS offsets of the two source columns and their weights (zero
    // when the tap falls outside the upsampled image: the convolution's SAME padding)
    int co[3][2];
    float wx[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int qx = x + d - 1;
        const bool vx = qx >= 0 && qx < p.W;
        const float sx = (float)min(max(qx, 0), p.W - 1) * p.wscale;
        const int c0 = (int)floorf(sx), c1 = min(c0 + 1, p.w - 1);
        const float lx = sx - (float)c0;
        co[d][0] = c0 * 18 + half;
        co[d][1] = c1 * 18 + half;
        wx[d][0] = vx ? 1.f - lx : 0.f;
        wx[d][1] = vx ? lx : 0.f;
    }
    const float4 bias4 = p.bias ? *reinterpret_cast<const float4 *>(p.bias + n0 + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);

    const int nbands = (p.H + p.RB - 1) / p.RB;
    int lo, hi;
    band_rows(0, lo, hi);
    {  // the first band's rows
        const int npix = (hi - lo + 1) * p.w;
        int row = lrow0, col = lcol0;
        for (int first = 0; first < npix; first += kPf * lgroups) {
            issue(lo, npix, first);
            commit(lo, npix, first, row, col);
        }
    }
    __syncthreads();
#if UPC_DEFER
    // r06 A/B (-DUPC_DEFER=1; NOT the shipped form): a band's results wait in registers (kGrp items per thread) and leave
    // AFTER the next band's rows have been committed.  Motive: gfx950 has one in-order counter for loads and stores, and
    // with a store inside the item loop hipcc's wait for it also drains the prefetch loads issued at the top of the band;
    // the knock-out builds (tools/gather_bench.py) showed the launch taking the SUM of its phases -- conv2_1 174 us = 51
    // (stores, set-up) + 70 (the 36 LDS reads per output: 126 B/clk/CU, the LDS pipe's rate) + 43 (z loads).  Measured
    // in one session against the r05 form: 243 vs 176 us (conv2_1), 389 vs 314 us (conv3_1) -- the stores as one burst
    // behind the band cost more than the loads they stopped draining; the r05 form stays.
    for (int k = 0; k < nbands; ++k) {
        // rows the next band adds: (hi, nhi] -- at most kPf * lgroups pixels (gather_geometry)
        int nlo = 0, nhi = hi;
        if (k + 1 < nbands) band_rows(k + 1, nlo, nhi);
        const int nnew = (nhi - hi) * p.w;
        issue(hi + 1, nnew, 0);
        const int y0 = k * p.RB, ylast = min(y0 + p.RB, p.H) - 1;
        float4 res[kGrp];
        auto flush = [&](int ybase) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < kGrp; ++g) {
                const int y = ybase + g * rpp;
                if (y > ylast) continue;
                float *o = OUT_C8 ? p.y + ((((size_t)b * (p.N / 8) + blk) * p.H + y) * p.W + x) * 8 + 4 * half
                                  : p.y + (((size_t)b * p.H + y) * p.W + x) * p.N + n0 + 4 * half;
#ifdef UPC_SKIP_STORE  // (timing experiments only)
                asm volatile("" ::"v"(res[g].x), "v"(res[g].y), "v"(res[g].z), "v"(res[g].w), "v"(o));
#elif UPC_NT & 2
                __builtin_nontemporal_store(f32x4{res[g].x, res[g].y, res[g].z, res[g].w}, reinterpret_cast<f32x4 *>(o));
#else
                *reinterpret_cast<float4 *>(o) = res[g];
#endif
            }
        };
        const int yb = y0 + rr;  // this thread's rows of the band: yb, yb + rpp, ... (at most kGrp: gather_geometry)
        {
#pragma unroll
            for (int g = 0; g < kGrp; ++g) {
                const int y = yb + g * rpp;
                if (y > ylast) continue;
                int ro[3][2];
                float wy[3][2];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int qy = y + d - 1;
                    const bool vy = qy >= 0 && qy < p.H;
                    const float sy = (float)min(max(qy, 0), p.H - 1) * p.hscale;
                    const int r0 = (int)floorf(sy), r1 = min(r0 + 1, p.h - 1);
                    const float ly = sy - (float)r0;
                    ro[d][0] = slot(r0) * rowf4;
                    ro[d][1] = slot(r1) * rowf4;
                    wy[d][0] = vy ? 1.f - ly : 0.f;
                    wy[d][1] = vy ? ly : 0.f;
                }
                f32x2 a01 = {bias4.x, bias4.y}, a23 = {bias4.z, bias4.w};
#ifdef UPC_SKIP_SUM  // (timing experiments only: one LDS read instead of 36, no arithmetic)
                {
                    const float4 v = src4[ro[1][0] + co[1][0]];
                    a01 += f32x2{v.x, v.y} * f32x2{wy[1][0], wx[1][0]};
                }
#else
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
                                const f32x2 w2 = {wgt, wgt};
                                a01 = __builtin_elementwise_fma(w2, f32x2{v.x, v.y}, a01);
                                a23 = __builtin_elementwise_fma(w2, f32x2{v.z, v.w}, a23);
                            }
                    }
#endif
                float4 acc = make_float4(a01.x, a01.y, a23.x, a23.y);
                if (p.relu) {
                    acc.x = fmaxf(acc.x, 0.f);
                    acc.y = fmaxf(acc.y, 0.f);
                    acc.z = fmaxf(acc.z, 0.f);
                    acc.w = fmaxf(acc.w, 0.f);
                }
                res[g] = acc;
            }
        }
        if (k + 1 == nbands) {
            flush(yb);
            break;
        }
        __syncthreads();  // every thread has finished reading this band's rows: the ring may be overwritten
        {
            int row = lrow0, col = lcol0;
            commit(hi + 1, nnew, 0, row, col);
        }
        flush(yb);  // behind the commit's wait for the rows: the stores no longer share a wait with the loads
        __syncthreads();
        hi = nhi;
    }
#else  // (the r05 form: every item stores its result at once -- kept for same-session A/B builds, -DUPC_DEFER=0)
    for (int k = 0; k < nbands; ++k) {
        // rows the next band adds: (hi, nhi] -- at most kPf * lgroups pixels (gather_geometry)
        int nlo = 0, nhi = hi;
        if (k + 1 < nbands) band_rows(k + 1, nlo, nhi);
        const int nnew = (nhi - hi) * p.w;
        issue(hi + 1, nnew, 0);
        const int y0 = k * p.RB, ylast = min(y0 + p.RB, p.H) - 1;
        for (int y = y0 + rr; y <= ylast; y += rpp) {
            int ro[3][2];
            float wy[3][2];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int qy = y + d - 1;
                const bool vy = qy >= 0 && qy < p.H;
                const float sy = (float)min(max(qy, 0), p.H - 1) * p.hscale;
                const int r0 = (int)floorf(sy), r1 = min(r0 + 1, p.h - 1);
                const float ly = sy - (float)r0;
                ro[d][0] = slot(r0) * rowf4;
                ro[d][1] = slot(r1) * rowf4;
                wy[d][0] = vy ? 1.f - ly : 0.f;
                wy[d][1] = vy ? ly : 0.f;
            }
            f32x2 a01 = {bias4.x, bias4.y}, a23 = {bias4.z, bias4.w};
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
                            const f32x2 w2 = {wgt, wgt};
                            a01 = __builtin_elementwise_fma(w2, f32x2{v.x, v.y}, a01);
                            a23 = __builtin_elementwise_fma(w2, f32x2{v.z, v.w}, a23);
                        }
                }
            float4 acc = make_float4(a01.x, a01.y, a23.x, a23.y);
            if (p.relu) {
                acc.x = fmaxf(acc.x, 0.f);
                acc.y = fmaxf(acc.y, 0.f);
                acc.z = fmaxf(acc.z, 0.f);
                acc.w = fmaxf(acc.w, 0.f);
            }
            float *o = OUT_C8 ? p.y + ((((size_t)b * (p.N / 8) + blk) * p.H + y) * p.W + x) * 8 + 4 * half
                              : p.y + (((size_t)b * p.H + y) * p.W + x) * p.N + n0 + 4 * half;
#if UPC_NT & 2
            __builtin_nontemporal_store(f32x4{acc.x, acc.y, acc.z, acc.w}, reinterpret_cast<f32x4 *>(o));
#else
            *reinterpret_cast<float4 *>(o) = acc;
#endif
        }
        if (k + 1 == nbands) break;
        __syncthreads();  // every thread has finished reading this band's rows: the ring may be overwritten
        {
            int row = lrow0, col = lcol0;
            commit(hi + 1, nnew, 0, row, col);
        }
        __syncthreads();
        hi = nhi;
    }
#endif
}

// ------------------------------------------------------------------------------------------------ backward
// dz = gather^T(dy): dz[s][t][n] = sum over the upsampled pixels q that interpolate from source pixel s of
// a[q][s] * dy[q - t][n] (q - t inside the output) -- the transpose of upconv_gather_kernel, written as a gather so that
// nothing is scattered and the sum order is fixed.  One workgroup per (32-channel group, source row, image): the dy rows
// that source row reaches (the upsampled rows that interpolate from it, one more on each side for the taps) sit in LDS
// as [row][x][32 channels] -- 128 contiguous bytes per pixel of the NHWC gradient, whole lines (a first version with one
// workgroup per (image, 8-channel block) read 32 of every 128 bytes and took 1.06 ms on conv3_1) -- next to the
// (upsampled index, weight) pairs of this source row and of every source column.  Item = (source column, tap, channel quad).
constexpr int kGtMax = 8;  // most upsampled rows (columns) that interpolate from one source row (column)
constexpr int kGtPf = 6;   // float4 registers per thread that carry the rows a step adds on their way into the ring
constexpr int kGtItemsA = 6, kGtItemsB = 8;  // most items per thread of the two passes of a step (3 W 8 / 256, 72 w / 256)

struct UpcGradParams {
    const float *dy;
    float *dz;
    int B, h, w, H, W, N, rows_max, cap;  // cap: rows of the ring, a power of two >= rows_max
    float hscale, wscale;
};

// One workgroup per (32-channel group, image) WALKS the source rows: the dy rows a source row reaches (the upsampled rows
// that interpolate from it, one more on each side for the taps) live in a ring of `cap` rows in LDS, [row][x][32 channels]
// = whole 128-byte lines of the NHWC gradient; a step adds the (about `scale`) rows its successor needs -- requested into
// registers before the step's arithmetic, written to the ring after it -- so every dy row is fetched ONCE (the first form,
// one workgroup per source row, fetched each ~3.5 times and waited for its whole window in front of 4 us of arithmetic:
// 0.58 ms on conv3_1; r04).  Per step, separable: V[dy][x] = sum_ky wy[ky] dy[yq[ky] - dy][x] for the three tap rows, then
// dz[sy][sx][t] = sum_kx wx[kx] V[dy(t)][xq[kx] - dx(t)].  The (index, weight) tables of every source row and column are
// made once per workgroup by the forward kernel's own arithmetic.
__global__ __launch_bounds__(256) void upconv_gather_t_kernel(const UpcGradParams p)
{
    extern __shared__ __attribute__((aligned(16))) float4 g4[];  // ring [cap][W][8] | V [3][W][8] | tables
    const int tid = threadIdx.x;
    const int grp = blockIdx.x, b = blockIdx.y;
    const int rowq = p.W * 8;  // float4 per dy row of this channel group
    float4 *V = g4 + (size_t)p.cap * rowq;
    int *xq = reinterpret_cast<int *>(V + (size_t)3 * rowq);  // [w][kGtMax]
    float *xw = reinterpret_cast<float *>(xq + p.w * kGtMax);
    int *xc = reinterpret_cast<int *>(xw + p.w * kGtMax);     // [w]
    int *yq = xc + ((p.w + 3) & ~3);                           // [h][kGtMax] (16-byte aligned: read as int4)
    float *yw = reinterpret_cast<float *>(yq + p.h * kGtMax);
    int *yc = reinterpret_cast<int *>(yw + p.h * kGtMax);     // [h]
    int *ylo = yc + p.h, *yhi = ylo + p.h;                     // [h]: first / last dy row a source row reaches
    // (upsampled index, weight) pairs: q interpolates from s iff floor(q scale) is s - 1 or s -- only q in
    // [(s - 1) / scale, (s + 1) / scale] need a look.  Thread t < h: source row t; h <= t < h + w: source column t - h.
    for (int t = tid; t < p.h + p.w; t += 256) {
        const bool row = t < p.h;
        const int sv = row ? t : t - p.h, n_in = row ? p.h : p.w, n_out = row ? p.H : p.W;
        const float scale = row ? p.hscale : p.wscale;
        const int q_lo = max((int)floorf((float)(sv - 1) / scale) - 1, 0), q_hi = min((int)ceilf((float)(sv + 1) / scale) + 1, n_out - 1);
        int *oq = row ? yq + sv * kGtMax : xq + sv * kGtMax;
        float *ow = row ? yw + sv * kGtMax : xw + sv * kGtMax;
        int c = 0, lo = n_out, hi = -1;
        for (int q = q_lo; q <= q_hi; ++q) {
            const float sq = (float)q * scale;
            const int r0 = (int)floorf(sq), r1 = min(r0 + 1, n_in - 1);
            const float l = sq - (float)r0;
            if ((r0 == sv || r1 == sv) && c < kGtMax) {
                oq[c] = q;
                ow[c] = (r0 == sv ? 1.f - l : 0.f) + (r1 == sv ? l : 0.f);
                ++c;
                lo = min(lo, q);
                hi = max(hi, q);
            }
        }
        if (row) {
            yc[sv] = c;
            ylo[sv] = c ? max(lo - 1, 0) : 0;
            yhi[sv] = c ? min(hi + 1, p.H - 1) : -1;
        } else {
            xc[sv] = c;
        }
    }
    __syncthreads();
    const float *src = p.dy + ((size_t)b * p.H * p.W) * p.N + grp * 32;  // this image, this channel group
    const int mask = p.cap - 1;
    f32x4 pf[kGtPf];
    // rows [r0, r1] of dy: requests into the registers (items tid, tid + 256, ...), then the ring
    auto request = [&](int r0, int r1) __attribute__((always_inline)) {
        const int total = (r1 - r0 + 1) * rowq;
#pragma unroll
        for (int j = 0; j < kGtPf; ++j) {
            const int i = j * 256 + tid;
            const int r = r0 + i / rowq, c = i - (i / rowq) * rowq;
            pf[j] = *reinterpret_cast<const f32x4 *>(i < total ? src + ((size_t)r * p.W + (c >> 3)) * p.N + 4 * (c & 7) : src);
        }
    };
    auto commit = [&](int r0, int r1) __attribute__((always_inline)) {
        const int total = (r1 - r0 + 1) * rowq;
#pragma unroll
        for (int j = 0; j < kGtPf; ++j) {
            const int i = j * 256 + tid;
            const int r = r0 + i / rowq, c = i - (i / rowq) * rowq;
            if (i < total) reinterpret_cast<f32x4 *>(g4)[(size_t)(r & mask) * rowq + c] = pf[j];
        }
    };
    // the first source row's window, kGtPf * 256 float4 at a time
    int have = -1;  // last dy row in the ring
    {
        int hi0 = -1;
        for (int sy = 0; sy < p.h && hi0 < 0; ++sy) hi0 = yhi[sy];  // (a source row that nothing interpolates from has no window)
        const int lo0 = 0;
        const int rows_per = max(1, (kGtPf * 256) / rowq);
        for (int r = lo0; r <= hi0; r += rows_per) {
            const int re = min(hi0, r + rows_per - 1);
            request(r, re);
            commit(r, re);
        }
        have = hi0;
    }
    __syncthreads();
    float *dzimg = p.dz + ((size_t)b * p.h * p.w) * 9 * p.N + (size_t)grp * 4 * 72;
    // a thread's items are the same for every source row: decoded once (the divisions were a third of the kernel's
    // instructions -- it is bound by its vector instructions, not by memory)
    int a_d[kGtItemsA], a_r[kGtItemsA];
#pragma unroll
    for (int k = 0; k < kGtItemsA; ++k) {
        const int it = tid + 256 * k;
        a_d[k] = it < 3 * rowq ? it / rowq : -1;
        a_r[k] = it - max(a_d[k], 0) * rowq;  // x * 8 + quad
    }
    int b_sx[kGtItemsB], b_v[kGtItemsB], b_dx[kGtItemsB], b_out[kGtItemsB];
#pragma unroll
    for (int k = 0; k < kGtItemsB; ++k) {
        const int it = tid + 256 * k;
        // (m = the float4's place in the 1152 contiguous bytes of (source pixel, channel group): consecutive lanes
        // write consecutive pieces -- a wave's store is eight whole lines)
        const int sx = it / 72, m = it - sx * 72, blk = m / 18, rem = m - 18 * blk, t = rem >> 1, quad = 2 * blk + (rem & 1);
        const int d = t / 3;
        b_sx[k] = it < p.w * 72 ? sx : -1;
        b_v[k] = d * rowq + quad;
        b_dx[k] = t - d * 3 - 1;
        b_out[k] = sx * 9 * p.N + 4 * m;
    }
    for (int sy = 0; sy < p.h; ++sy) {
        // the rows the NEXT source row adds: requested now, in the ring after this row's arithmetic
        const int next_hi = sy + 1 < p.h ? max(yhi[sy + 1], have) : have;
        const bool more = next_hi > have;
        if (more) request(have + 1, next_hi);
        // (the first four (index, weight) pairs of a row / column as one 16-byte read each and branch-free -- a pair
        // that does not apply gets weight 0 and a clamped address --, so that an item is two LDS round trips deep
        // instead of two per pair: the kernel was bound by those dependent reads at two waves per SIMD)
        const int ny = yc[sy];
        const int4 yq4 = *reinterpret_cast<const int4 *>(yq + sy * kGtMax);
        const float4 yw4 = *reinterpret_cast<const float4 *>(yw + sy * kGtMax);
#pragma unroll
        for (int k = 0; k < kGtItemsA; ++k) {
            const int d = a_d[k], r = a_r[k], it = tid + 256 * k;
            if (d < 0) continue;
            const int py0 = yq4.x - (d - 1), py1 = yq4.y - (d - 1), py2 = yq4.z - (d - 1), py3 = yq4.w - (d - 1);
            const float w0 = (ny > 0 && py0 >= 0 && py0 < p.H) ? yw4.x : 0.f, w1 = (ny > 1 && py1 >= 0 && py1 < p.H) ? yw4.y : 0.f;
            const float w2 = (ny > 2 && py2 >= 0 && py2 < p.H) ? yw4.z : 0.f, w3 = (ny > 3 && py3 >= 0 && py3 < p.H) ? yw4.w : 0.f;
            const float4 v0 = g4[(size_t)(py0 & mask) * rowq + r], v1 = g4[(size_t)(py1 & mask) * rowq + r];
            const float4 v2 = g4[(size_t)(py2 & mask) * rowq + r], v3 = g4[(size_t)(py3 & mask) * rowq + r];
            f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
            auto acc = [&](float wv, const float4 &v) __attribute__((always_inline)) {
                const f32x2 w2_ = {wv, wv};
                a01 = __builtin_elementwise_fma(w2_, f32x2{v.x, v.y}, a01);
                a23 = __builtin_elementwise_fma(w2_, f32x2{v.z, v.w}, a23);
            };
            // (a slot whose weight is 0 may hold anything, NaN included: select, do not multiply)
            if (w0 != 0.f) acc(w0, v0);
            if (w1 != 0.f) acc(w1, v1);
            if (w2 != 0.f) acc(w2, v2);
            if (w3 != 0.f) acc(w3, v3);
            for (int ky = 4; ky < ny; ++ky) {
                const int py = yq[sy * kGtMax + ky] - (d - 1);
                if (py < 0 || py >= p.H) continue;
                acc(yw[sy * kGtMax + ky], g4[(size_t)(py & mask) * rowq + r]);
            }
            V[it] = make_float4(a01.x, a01.y, a23.x, a23.y);
        }
        __syncthreads();  // V complete; every read of the ring for this source row is done
        float *dzrow = dzimg + (size_t)sy * p.w * 9 * p.N;
#pragma unroll
        for (int k = 0; k < kGtItemsB; ++k) {  // item = (source column, tap, channel quad)
            const int sx = b_sx[k], dx = b_dx[k];
            if (sx < 0) continue;
            const int nx = xc[sx];
            const int4 q4 = *reinterpret_cast<const int4 *>(xq + sx * kGtMax);
            const float4 w4 = *reinterpret_cast<const float4 *>(xw + sx * kGtMax);
            const float4 *row = V + b_v[k];
            const int px0 = q4.x - dx, px1 = q4.y - dx, px2 = q4.z - dx, px3 = q4.w - dx;
            const bool ok0 = nx > 0 && px0 >= 0 && px0 < p.W, ok1 = nx > 1 && px1 >= 0 && px1 < p.W;
            const bool ok2 = nx > 2 && px2 >= 0 && px2 < p.W, ok3 = nx > 3 && px3 >= 0 && px3 < p.W;
            const float4 v0 = row[(ok0 ? px0 : 0) * 8], v1 = row[(ok1 ? px1 : 0) * 8];
            const float4 v2 = row[(ok2 ? px2 : 0) * 8], v3 = row[(ok3 ? px3 : 0) * 8];
            f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
            auto acc = [&](float wv, const float4 &v) __attribute__((always_inline)) {
                const f32x2 w2_ = {wv, wv};
                a01 = __builtin_elementwise_fma(w2_, f32x2{v.x, v.y}, a01);
                a23 = __builtin_elementwise_fma(w2_, f32x2{v.z, v.w}, a23);
            };
            if (ok0) acc(w4.x, v0);
            if (ok1) acc(w4.y, v1);
            if (ok2) acc(w4.z, v2);
            if (ok3) acc(w4.w, v3);
            for (int kx = 4; kx < nx; ++kx) {
                const int px = xq[sx * kGtMax + kx] - dx;
                if (px < 0 || px >= p.W) continue;
                acc(xw[sx * kGtMax + kx], row[px * 8]);
            }
            *reinterpret_cast<float4 *>(dzrow + b_out[k]) = make_float4(a01.x, a01.y, a23.x, a23.y);
        }
        // (as late as possible: the requests have had the whole step to land; overwrites rows below the next window --
        // cap >= the widest window -- that nobody reads any more since the barrier above)
        if (more) commit(have + 1, next_hi);
        have = next_hi;
        __syncthreads();  // the ring holds the next window; V may be overwritten
    }
}

// wt[c][j] = g[n][t*C + c], j = (n / 8) * 72 + t * 8 + (n % 8): the (C, 9 N) weight matrix of the data-gradient GEMM
__global__ __launch_bounds__(256) void upconv_weights_t_kernel(const float *__restrict__ g, int N, int C, float *__restrict__ wt)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int K = 9 * N;
    if (i >= (long long)C * K) return;
    const int c = (int)(i / K), j = (int)(i - (long long)c * K);
    const int blk = j / 72, r = j - blk * 72, t = r >> 3, n = blk * 8 + (r & 7);
    wt[i] = g[((size_t)n * 9 + t) * C + c];
}

// dw[n][t*C + c] += dwp[j][c]: folds the tap GEMM's weight gradient back into the filter's layout
__global__ __launch_bounds__(256) void upconv_fold_dw_kernel(const float *__restrict__ dwp, int N, int C, float *__restrict__ dw)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C / 4;
    if (i >= (long long)9 * N * c4) return;
    const int j = (int)(i / c4), q = (int)(i - (long long)j * c4);
    const int blk = j / 72, r = j - blk * 72, t = r >> 3, n = blk * 8 + (r & 7);
    float4 *d = reinterpret_cast<float4 *>(dw + ((size_t)n * 9 + t) * C) + q;
    const float4 a = *d, v = reinterpret_cast<const float4 *>(dwp)[i];
    *d = make_float4(a.x + v.x, a.y + v.y, a.z + v.z, a.w + v.w);
}

// Launch geometry of the gather: threads = rpp x 2 W (rows of a pass x (column, channel half)), rpp a power of two with
// at most 512 threads; RB = output rows per band (8, or rpp if larger); cap = the ring's rows, a power of two >= the rows
// any band reaches; the rows a band ADDS must fit the prefetch registers.  false: the map is too wide for this kernel.
bool gather_geometry(int h, int w, int H, int W, float hscale, int *threads, int *RB, int *cap)
{
    if (2 * W > 512 || w < 1) return false;
    int rpp = 1;
    while (rpp * 2 <= 16 && rpp * 2 * 2 * W <= 512) rpp *= 2;
    const int nthr = rpp * 2 * W;
    if (nthr < 18) return false;
    // a map whose source rows all fit a third of the LDS is ONE band (three workgroups per CU overlap each other's load
    // and arithmetic phases); otherwise bands of 8 rows (or one pass) with the next band's rows prefetched
    int rb = rpp > 8 ? rpp : 8;
    const int max_rb = UPC_DEFER ? kGrp * rpp : 1 << 30;  // (the deferred-store A/B form keeps kGrp rows per thread in registers)
    if (rb > max_rb) rb = max_rb;
    if (g_upconv_band.load() > 0) {  // (tuning knob: bands of this many rows, also where one band would do)
        rb = (g_upconv_band.load() + rpp - 1) / rpp * rpp;
        if (rb > max_rb) rb = max_rb;
    } else if ((size_t)h * w * 288 <= kGatherLdsBytes * 3 / 5 && (H + rpp - 1) / rpp * rpp <= max_rb) {
        rb = (H + rpp - 1) / rpp * rpp;
    }
    // the kernel's own band arithmetic: the most rows a band reaches, the most a further band adds
    int rows = 0, add = 0, prev_hi = -1;
    for (int k = 0; k * rb < H; ++k) {
        const int y0 = k * rb, yl = (y0 + rb < H ? y0 + rb : H) - 1;
        const int lo = (int)floorf((float)(y0 - 1 > 0 ? y0 - 1 : 0) * hscale);
        int hi = (int)floorf((float)(yl + 1 < H - 1 ? yl + 1 : H - 1) * hscale) + 1;
        if (hi > h - 1) hi = h - 1;
        if (hi - lo + 1 > rows) rows = hi - lo + 1;
        if (k > 0 && hi - prev_hi > add) add = hi - prev_hi;
        prev_hi = hi;
    }
    if (add * w > kPf * (nthr / 18)) return false;  // the rows a band adds travel through the prefetch registers
    const int c = rows;
    if ((size_t)c * w * 288 > kGatherLdsBytes) return false;
    *threads = nthr;
    *RB = rb;
    *cap = c;
    return true;
}

}  // namespace

// images per GEMM + gather round, as MB of z (0 = the whole batch at once); see conv3x3_upsampled
static std::atomic<int> g_upconv_chunk_mb{0};
extern "C" void mpsr_debug_set_upconv_chunk_mb(int mb) { g_upconv_chunk_mb = mb > 0 ? mb : 0; }
extern "C" void mpsr_debug_set_upconv_band(int rows) { g_upconv_band = rows > 0 ? rows : 0; }

namespace mpsr {

static bool upconv_geometry_ok(int h, int w, int OH, int OW, int align_corners)
{
    int t, rb, cap;
    const float hs = (align_corners && OH > 1) ? (float)(h - 1) / (float)(OH - 1) : (float)h / (float)OH;
    return gather_geometry(h, w, OH, OW, hs, &t, &rb, &cap);
}

size_t upconv_weight_floats(int C, int N) { return (size_t)9 * N * C; }
size_t upconv_z_floats(long long Msrc, int N) { return (size_t)Msrc * 9 * (size_t)N; }

// x (B,h,w,C) NHWC, 3x3 filter (N, 9 C), output (B,OH,OW,N): N a multiple of 128 (whole GEMM parts), the source map a
// shape the pointwise kernel takes, 32-bit byte offsets inside z parts
bool upconv_applies(int B, int h, int w, int C, int OH, int OW, int N, int align_corners)
{
    const long long M = (long long)B * h * w;
    return B > 0 && h >= 1 && w >= 1 && OH >= 1 && OW >= 1 && N >= 128 && N % 128 == 0 && C % 4 == 0 &&
           pointwise_applies(M, C, PART) && upconv_geometry_ok(h, w, OH, OW, align_corners) && B <= 65535 &&
           (long long)B * OH * OW * N * 4 < 0x7fffffffffLL;
}

int conv3x3_upsampled(const float *x, int B, int h, int w, int C, int OH, int OW, int align_corners, const float *g,
                      const float *bias, int relu, float *y, int N, int out_c8, float *z, size_t z_floats, float *ws,
                      size_t ws_floats, hipStream_t s)
{
    MPSR_REQUIRE(upconv_applies(B, h, w, C, OH, OW, N, align_corners), "conv3x3_upsampled: unsupported shape (B=%d %dx%dx%d -> %dx%dx%d)",
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
        ready = g_filter_cache_slot.holds(FILTER_FORM_UPCONV);
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
    UpcParams p;
    p.bias = bias;
    p.part_stride = (size_t)M * PART;
    p.h = h; p.w = w; p.H = OH; p.W = OW; p.N = N; p.relu = relu;
    p.hscale = (align_corners && OH > 1) ? (float)(h - 1) / (float)(OH - 1) : (float)h / (float)OH;
    p.wscale = (align_corners && OW > 1) ? (float)(w - 1) / (float)(OW - 1) : (float)w / (float)OW;
    int threads = 0;
    if (!gather_geometry(h, w, OH, OW, p.hscale, &threads, &p.RB, &p.cap))
        return fail(MPSR_ERR_UNSUPPORTED, "conv3x3_upsampled: no gather geometry for a %dx%d -> %dx%d map", h, w, OH, OW);
    p.capdiv = make_fastdiv(p.cap);
    const size_t lds = (size_t)p.cap * w * 288;
    // Images in chunks: GEMM of a chunk, then its gather -- with the chunk's z (images x h w x 9 N floats) no larger than
    // the Infinity Cache holds next to the other operands, the gather finds most of what the GEMM just wrote still on
    // the die instead of in HBM (g_upconv_chunk_mb; 0 = the whole batch at once)
    const int chunk_mb = g_upconv_chunk_mb.load();
    int bc = B;
    if (chunk_mb > 0) {
        const double per_image = (double)h * w * 9.0 * N * 4.0;
        bc = (int)((double)chunk_mb * 1048576.0 / per_image);
        bc = bc < 8 ? 8 : bc / 8 * 8;
        const int nch = ceil_div(B, bc);
        bc = ceil_div(ceil_div(B, nch), 8) * 8;  // even chunks, whole groups of 8 images (one per XCD)
    }
    for (int b0 = 0; b0 < B; b0 += bc) {
        const int nb = B - b0 < bc ? B - b0 : bc;
        const size_t pix0 = (size_t)b0 * h * w;
        for (int pidx = 0; pidx < parts; ++pidx) {
            const int rc = conv1x1_pointwise(x + pix0 * C, (long long)nb * h * w, C, wp + (size_t)pidx * PART * C, nullptr,
                                             nullptr, 0, z + (size_t)pidx * M * PART + pix0 * PART, PART, s);
            if (rc) return rc;
        }
        p.z = z + pix0 * PART;
        p.y = y + (size_t)b0 * OH * OW * N;
        p.B = nb;
        const dim3 grid((unsigned)(8 * ceil_div(nb, 8) * (N / 8)));
        if (out_c8) {
            MPSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(upconv_gather_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(upconv_gather_kernel<true>, grid, dim3(threads), lds, s, p);
        } else {
            MPSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(upconv_gather_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(upconv_gather_kernel<false>, grid, dim3(threads), lds, s, p);
        }
        MPSR_CHECK_LAUNCH("upconv_gather_kernel");
    }
    return MPSR_OK;
}

}  // namespace mpsr

namespace mpsr {

// most dy rows one source row reaches in upconv_gather_t_kernel: the upsampled rows that interpolate from it + one on
// each side (the kernel's own arithmetic)
static int gather_t_rows(int h, int OH, float hscale)
{
    int most = 1;
    for (int sy = 0; sy < h; ++sy) {
        int lo = OH, hi = -1;
        for (int q = 0; q < OH; ++q) {
            const float sq = (float)q * hscale;
            const int r0 = (int)floorf(sq), r1 = r0 + 1 < h - 1 ? r0 + 1 : h - 1;
            if (r0 == sy || r1 == sy) {
                lo = q < lo ? q : lo;
                hi = q > hi ? q : hi;
            }
        }
        if (hi >= lo) {
            const int a = lo - 1 > 0 ? lo - 1 : 0, b = hi + 1 < OH - 1 ? hi + 1 : OH - 1;
            most = b - a + 1 > most ? b - a + 1 : most;
        }
    }
    return most;
}
static int gather_t_cap(int rows)
{
    int cap = 1;
    while (cap < rows) cap *= 2;
    return cap;
}
// LDS of upconv_gather_t_kernel: the ring (cap rows x 32 channels), the three row-reduced rows V, the row / column tables
static size_t gather_t_lds_bytes(int rows, int h, int w, int OW)
{
    return (size_t)(gather_t_cap(rows) + 3) * OW * 128 + (size_t)w * (kGtMax * 8 + 4) + 16 + (size_t)h * (kGtMax * 8 + 12);
}
// most dy rows a step adds (the next source row's window beyond this one's)
static int gather_t_new_rows(int h, int OH, float hscale)
{
    int most = 0, have = -1;
    for (int sy = 0; sy < h; ++sy) {
        int hi = -1;
        for (int q = 0; q < OH; ++q) {
            const float sq = (float)q * hscale;
            const int r0 = (int)floorf(sq), r1 = r0 + 1 < h - 1 ? r0 + 1 : h - 1;
            if (r0 == sy || r1 == sy) hi = q;
        }
        if (hi >= 0) {
            const int e = hi + 1 < OH - 1 ? hi + 1 : OH - 1;
            if (have >= 0 && e - have > most) most = e - have;
            if (e > have) have = e;
        }
    }
    return most;
}

bool upconv_bwd_applies(int B, int h, int w, int C, int OH, int OW, int N, int align_corners)
{
    const long long M = (long long)B * h * w;
    // (at most kGtMax upsampled rows per source row: scales down to 1/3)
    const bool scale_ok = (long long)OH <= 3LL * h && (long long)OW <= 3LL * w;
    const float hs = (align_corners && OH > 1) ? (float)(h - 1) / (float)(OH - 1) : (float)h / (float)OH;
    return upconv_applies(B, h, w, C, OH, OW, N, align_corners) && scale_ok && OH <= 65535 && h <= 65535 &&
           gather_t_lds_bytes(gather_t_rows(h, OH, hs), h, w, OW) <= 80 * 1024 &&
           gather_t_new_rows(h, OH, hs) * OW * 8 <= kGtPf * 256 && 3 * OW * 8 <= kGtItemsA * 256 && 72 * w <= kGtItemsB * 256 &&
           pointwise_applies(M, 9 * N, C) &&
           M * 9 * N * 4 < 0xfffff000LL && M < (1LL << 24);
}

// floats: dz (M x 9N) | dW' (9N x C) | W'^T (C x 9N)
size_t upconv_bwd_scratch_floats(int B, int h, int w, int C, int N)
{
    const size_t M = (size_t)B * h * w;
    return align_up(M * 9 * N, 64) + 2 * align_up((size_t)9 * N * C, 64);
}

}  // namespace mpsr

extern "C" int mpsr_conv2d_wgrad_f32(const float *x, const float *dy, int B, int H, int W, int C, int N, int KH, int KW,
                                     int dilation, float *dw, float *db, mpsr_stream_t stream);

namespace mpsr {

// Backward of conv3x3_upsampled (bias-free part): dy (B,OH,OW,N) NHWC -> dw (N, 9 C) += weight gradient, dx (B,h,w,C) =
// data gradient (or nullptr).  dz = gather^T(dy) at the SOURCE resolution; dW' = dz^T x (a 1x1 weight gradient with 9 N
// outputs) folded back into the filter layout; dx = dz W' (a 1x1 GEMM with K = 9 N on the pointwise kernel) -- the
// resize gradient, the F(4x4) data gradient on the upsampled map and the Winograd-domain weight gradient in one.
int conv3x3_upsampled_bwd(const float *x, const float *dy, int B, int h, int w, int C, int OH, int OW, int align_corners,
                          const float *g, int N, float *dw, float *dx, float *ws, size_t ws_floats, hipStream_t s)
{
    MPSR_REQUIRE(upconv_bwd_applies(B, h, w, C, OH, OW, N, align_corners), "conv3x3_upsampled_bwd: unsupported shape");
    if (!ws || ws_floats < upconv_bwd_scratch_floats(B, h, w, C, N))
        return fail(MPSR_ERR_WORKSPACE, "conv3x3_upsampled_bwd: scratch holds %zu floats, needs %zu", ws_floats,
                    upconv_bwd_scratch_floats(B, h, w, C, N));
    const size_t M = (size_t)B * h * w, wf = align_up((size_t)9 * N * C, 64);
    float *dz = ws, *dwp = ws + align_up(M * 9 * N, 64), *wt = dwp + wf;
    UpcGradParams p;
    p.dy = dy; p.dz = dz;
    p.B = B; p.h = h; p.w = w; p.H = OH; p.W = OW; p.N = N;
    p.hscale = (align_corners && OH > 1) ? (float)(h - 1) / (float)(OH - 1) : (float)h / (float)OH;
    p.wscale = (align_corners && OW > 1) ? (float)(w - 1) / (float)(OW - 1) : (float)w / (float)OW;
    p.rows_max = gather_t_rows(h, OH, p.hscale);
    p.cap = gather_t_cap(p.rows_max);
    const size_t lds = gather_t_lds_bytes(p.rows_max, h, w, OW);
    MPSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(upconv_gather_t_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(upconv_gather_t_kernel, dim3((unsigned)(N / 32), (unsigned)B), dim3(256), lds, s, p);
    MPSR_CHECK_LAUNCH("upconv_gather_t_kernel");
    // weight gradient of the tap GEMM, folded back into (N, 9 C)
    MPSR_CHECK_HIP(hipMemsetAsync(dwp, 0, (size_t)9 * N * C * sizeof(float), s));
    if (int rc = mpsr_conv2d_wgrad_f32(x, dz, B, h, w, C, 9 * N, 1, 1, 1, dwp, nullptr, reinterpret_cast<mpsr_stream_t>(s))) return rc;
    {
        const long long total = (long long)9 * N * (C / 4);
        hipLaunchKernelGGL(upconv_fold_dw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dwp, N, C, dw);
        MPSR_CHECK_LAUNCH("upconv_fold_dw_kernel");
    }
    if (dx) {
        const long long total = (long long)C * 9 * N;
        hipLaunchKernelGGL(upconv_weights_t_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, N, C, wt);
        MPSR_CHECK_LAUNCH("upconv_weights_t_kernel");
        if (int rc = conv1x1_pointwise(dz, (long long)M, 9 * N, wt, nullptr, nullptr, 0, dx, C, s)) return rc;
    }
    return MPSR_OK;
}

}  // namespace mpsr

extern "C" int mpsr_conv3x3_upsampled_applies(int B, int h, int w, int C, int OH, int OW, int N, int align_corners)
{
    return mpsr::upconv_applies(B, h, w, C, OH, OW, N, align_corners) ? (mpsr::upconv_bwd_applies(B, h, w, C, OH, OW, N, align_corners) ? 2 : 1) : 0;
}

extern "C" size_t mpsr_conv3x3_upsampled_bwd_scratch_floats(int B, int h, int w, int C, int N)
{
    if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || N <= 0) return 0;
    return mpsr::upconv_bwd_scratch_floats(B, h, w, C, N);
}

extern "C" int mpsr_conv3x3_upsampled_bwd_f32(const float *x, const float *dy, int B, int h, int w, int C, int OH, int OW,
                                              int align_corners, const float *weights, int N, float *dw, float *dx,
                                              float *ws, size_t ws_floats, mpsr_stream_t stream)
{
    MPSR_REQUIRE(B >= 0 && h > 0 && w > 0 && C > 0 && OH > 0 && OW > 0 && N > 0, "conv3x3_upsampled_bwd: bad shape");
    if (B == 0) return MPSR_OK;
    MPSR_REQUIRE(x && dy && weights && dw && ws, "conv3x3_upsampled_bwd: null pointer");
    if (!mpsr::upconv_bwd_applies(B, h, w, C, OH, OW, N, align_corners))
        return mpsr::fail(MPSR_ERR_UNSUPPORTED, "conv3x3_upsampled_bwd: shape outside what the tap-GEMM backward takes "
                                                "(B=%d %dx%dx%d -> %dx%dx%d); use the resize / conv gradients", B, h, w, C, OH, OW, N);
    return mpsr::conv3x3_upsampled_bwd(x, dy, B, h, w, C, OH, OW, align_corners, weights, N, dw, dx, ws, ws_floats,
                                       mpsr::as_stream(stream));
}

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
    if (!mpsr::upconv_applies(B, h, w, C, OH, OW, N, align_corners))
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
