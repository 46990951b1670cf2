// 3x3 stride-1 SAME convolution by the Winograd minimal-filtering algorithm F(2x2, 3x3) on fp32 MFMA, for the map
// decoder's layers (24x24 / 48x48 maps, 128-512 channels: 32 % of the direct-convolution multiply-adds of a step).
//
//   y (2x2 block of output pixels) = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A,   d = the 4x4 input patch around it.
// 16 products per output pair of (channel, 2x2 block) instead of 36: the contraction shrinks 2.25x.  The 16 elementwise
// positions are 16 independent GEMMs  M_p[tile][n] = sum_c V_p[tile][c] * U_p[n][c]  (tile = one 2x2 output block).
//
// MI355X design: ONE kernel does input transform, the 16 GEMMs and the output transform -- neither V nor M ever reach
// HBM (as separate passes they would move 8x the layer's own bytes).
//   * workgroup = 64 tiles x 64 output channels x all 16 positions; 4 waves as 2 x 2, each wave 32 tiles x 32 channels
//     x 16 positions = 16 accumulator tiles of v_mfma_f32_32x32x2_f32 = 256 accumulator registers (one wave per SIMD:
//     the kernel is built for the 512-register budget that gives);
//   * per K step of 16 input channels every thread loads the 16 pixels of one tile's patch for 4 channels (16-byte
//     loads, out-of-image pixels by the buffer bounds check), transforms them in registers (32 add/sub per channel) and
//     writes the 16 positions' A rows to LDS; the transformed filters U (made once per call by wino_filter_kernel, laid
//     out [channel block][position][n][16] so a workgroup's slab is one contiguous 64 KiB read) go to LDS unchanged;
//     the loads of step k+1 are in flight while step k's 128 MFMAs per wave run;
//   * LDS rows are 16 + 4 floats: conflict-free ds_read_b128 fragments, one read feeds four MFMAs (same k-ordering
//     trick as conv_mfma.hip);
//   * the output transform is register arithmetic: all 16 position accumulators of a wave have the same lane layout,
//     so A^T M A is 24 adds per element; bias + ReLU fused; each store instruction writes two 128-byte rows.
// Numerics: fp32 throughout; transform constants are 0, +-1, +-1/2 (exact); measured error against float64
// ~1e-6 of the tensor scale (direct fp32 MFMA path: ~5e-7), three orders inside the path's 1e-3 budget.
#include <atomic>
#include <mutex>
#include <type_traits>

#include "common.h"
#include "wino3_filter.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int KC = 16;            // input channels per K step
constexpr int ROW = KC + 4;       // floats per LDS row
constexpr int MT = 64, NT = 64;   // tiles x output channels per workgroup
constexpr int kLdsFloats = 2 * 16 * 64 * ROW;

struct WinoParams {
    const float *x, *u, *bias;
    float *y;
    int B, H, W, C, N, th, tw, T;  // th x tw tiles per pixel sub-grid, T tiles in all
    // atrous layers: at dilation d the pixels (d i + a, d j + b) of one (a, b) form an independent dense image of
    // H / d x W / d pixels (its taps only reach its own pixels; outside it is SAME-padding zeros), so the layer is d^2
    // dense 3x3 convolutions per image on strided views: tile = (image, sub-grid (a, b), 2x2 block of the sub-grid).
    // (ResNet-101's block2 / block3 on the FULL image, 40 x 152 at dilation 2 / 4: reference graph resnet_v1.py:116-127,
    // resnet_utils.py:194-196; the 12 x 12 crop maps take the one-tile-per-sub-grid F(3x3,3x3) kernels instead.)
    int dil;
    int cblocks, nblocks, mblocks, relu;
    unsigned xbytes, ubytes;
    unsigned long long *trace;  // -DWINO_TRACE builds: per-workgroup timestamps (tools/wino_trace.py)
    // SPLIT instantiation of the eight-wave kernel (launches too small to fill the chip: one full image): K slices;
    // slice k runs channel steps [k * steps, (k + 1) * steps) and stores PARTIAL outputs into part + k * yfloats
    // (winograd3z.hip: winograd_finish_slices adds them in order, + bias, ReLU)
    float *part;
    int nslices, steps;
    size_t yfloats;
};

// U[cb][pos][n][KC] = (G g G^T)[pos] for filter g = w[n][(ky*3+kx)*C + c], c = cb*KC + j.  One thread per (n, c).
__global__ __launch_bounds__(256) void wino_filter_kernel(const float *__restrict__ w, int N, int C, float *__restrict__ u)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * C) return;
    const int n = (int)(i / C), c = (int)(i - (long long)n * C);
    double g[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = (double)w[(size_t)n * 9 * C + (size_t)(ky * 3 + kx) * C + c];
    double t[4][3];  // G g
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        t[0][kx] = g[0][kx];
        t[1][kx] = 0.5 * (g[0][kx] + g[1][kx] + g[2][kx]);
        t[2][kx] = 0.5 * (g[0][kx] - g[1][kx] + g[2][kx]);
        t[3][kx] = g[2][kx];
    }
    float *dst = u + ((size_t)(c / KC) * 16 * N + n) * KC + (c % KC);
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
        const double r0 = t[xi][0], r1 = 0.5 * (t[xi][0] + t[xi][1] + t[xi][2]),
                     r2 = 0.5 * (t[xi][0] - t[xi][1] + t[xi][2]), r3 = t[xi][2];
        dst[(size_t)(xi * 4 + 0) * N * KC] = (float)r0;
        dst[(size_t)(xi * 4 + 1) * N * KC] = (float)r1;
        dst[(size_t)(xi * 4 + 2) * N * KC] = (float)r2;
        dst[(size_t)(xi * 4 + 3) * N * KC] = (float)r3;
    }
}

// tile -> image, origin pixel of its sub-grid (a, b) and 2x2 block (ty, tx) of the sub-grid
struct WinoTile {
    int img, a, b, ty, tx;
};
__device__ __forceinline__ WinoTile wino_tile(const WinoParams &p, int t)
{
    WinoTile w;
    const int tps = p.th * p.tw, tpi = tps * p.dil * p.dil;
    w.img = t / tpi;
    const int rem = t - w.img * tpi, sub = rem / tps, tt = rem - sub * tps;
    w.a = sub / p.dil;
    w.b = sub - w.a * p.dil;
    w.ty = tt / p.tw;
    w.tx = tt - w.ty * p.tw;
    return w;
}

using f32x2 = __attribute__((ext_vector_type(2))) float;

// a float4 as two packed pairs: the transforms then issue as v_pk_add_f32 (two adds per VALU slot)
struct F4 {
    f32x2 lo, hi;
};
__device__ __forceinline__ F4 ld4(const float4 &v) { return F4{f32x2{v.x, v.y}, f32x2{v.z, v.w}}; }
__device__ __forceinline__ F4 operator+(const F4 &a, const F4 &b) { return F4{a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ F4 operator-(const F4 &a, const F4 &b) { return F4{a.lo - b.lo, a.hi - b.hi}; }
__device__ __forceinline__ void st4(float *dst, const F4 &v)
{
    *reinterpret_cast<float4 *>(dst) = make_float4(v.lo[0], v.lo[1], v.hi[0], v.hi[1]);
}

// K-loop schedule.  The 16 positions form two groups (g0 = positions 0-7, g1 = 8-15) with their own LDS regions.
// One wave per SIMD means nothing else covers the time a wave spends filling LDS, so the fill of the NEXT data is
// woven between the MFMAs of the CURRENT data, one 16-byte store per half step of 4 MFMAs:
//     first half  of step k:  multiply g0(k)   while storing g1(k)      | barrier
//     second half of step k:  multiply g1(k)   while storing g0(k+1)    | barrier
// (a region is rewritten only after the barrier that follows its last read, and read only after the barrier that
// follows its last store).  Global loads run a whole half step or more ahead of the store that consumes them; their
// addresses are per-thread constants plus a scalar offset, so a load costs no vector ALU work.
__global__ __launch_bounds__(256, 1) void wino_conv_kernel(const WinoParams p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;                   // [16][MT][ROW]
    float *Bs = lds + 16 * MT * ROW;   // [16][NT][ROW]

    // XCD x (workgroup b runs on XCD b % 8: speed only) takes M blocks x, x+8, ...; the N blocks of one M block run
    // back to back on it, so the patches are fetched from HBM once
    const int xcd = blockIdx.x & 7, l_ = blockIdx.x >> 3;
    const int nb = l_ % p.nblocks;
    const int mb = (l_ / p.nblocks) * 8 + xcd;
    if (mb >= p.mblocks) return;  // block-uniform
    const int n0 = nb * NT, t0 = mb * MT;
#ifdef WINO_TRACE
    unsigned long long ts[8];
    int nts = 0;
    ts[nts++] = __builtin_readcyclecounter();
#define WINO_STAMP() do { if (nts < 8) ts[nts++] = __builtin_readcyclecounter(); } while (0)
#else
#define WINO_STAMP() do { } while (0)
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, (int)p.ubytes, 0x00020000);

    // ---- A loader: thread = (tile, channel quad); the 4x4 patch starts one pixel up-left of the tile's 2x2 block.
    // One byte offset per patch pixel, fixed for the whole K loop (out-of-image pixels: an out-of-range offset, which
    // the buffer load answers with zeros); the channel block comes in through the scalar offset.
    const int ltile = tid >> 2, quad = tid & 3;
    unsigned aoff[16];
    {
        const int t = t0 + ltile;
        const WinoTile wt = wino_tile(p, t < p.T ? t : 0);
        const int y0 = 2 * wt.ty - 1, x0 = 2 * wt.tx - 1;  // sub-grid coordinates of the patch's corner
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool ok = t < p.T && y0 + r >= 0 && y0 + r < 2 * p.th && x0 + s >= 0 && x0 + s < 2 * p.tw;
                const int yy = wt.a + p.dil * (y0 + r), xx = wt.b + p.dil * (x0 + s);
                aoff[r * 4 + s] = ok ? ((unsigned)((wt.img * p.H + yy) * p.W + xx) * (unsigned)p.C + 4u * quad) * 4u
                                     : p.xbytes;
            }
    }
    // ---- B loader: position j's slab row (n, channel quad) of this thread; rows past N read as zeros
    const unsigned boff = n0 + (tid >> 2) < p.N ? (unsigned)((n0 + (tid >> 2)) * KC + 4 * (tid & 3)) * 4u : p.ubytes;
    const unsigned bstride = (unsigned)p.N * KC * 4u;  // bytes between positions (and 16 x that between channel blocks)

    float4 pa[16], pb[8];
    // One patch pixel.  `live` false (past the last channel block) turns the load into an out-of-range one (zeros, no
    // traffic); the loads are issued UNCONDITIONALLY: around a branch hipcc can no longer count them and makes every
    // later wait on the filter loads wait for these patch loads too.
    auto load_a1 = [&](int cb, bool live, int i) {
        pa[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                               rx, live ? aoff[i] : p.xbytes, live ? (unsigned)cb * KC * 4u : 0u, 0));
    };
    // One filter row (position pos of channel block cb) into staging slot `slot`: always an L2 hit (U is a few MB
    // shared by every workgroup), so it is fetched only about a third of a half step before its store.
    auto load_b1 = [&](int cb, int pos, int slot) {
        pb[slot] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  ru, boff, ((unsigned)cb * 16u + (unsigned)pos) * bstride, 0));
    };
    // rows of the input transform: t = B^T d, kept between the two halves of a step
    F4 t[4][4];
    auto row_transform_col = [&](int c) {
        const F4 d0 = ld4(pa[c]), d1 = ld4(pa[4 + c]), d2 = ld4(pa[8 + c]), d3 = ld4(pa[12 + c]);
        t[0][c] = d0 - d2;
        t[1][c] = d1 + d2;
        t[2][c] = d2 - d1;
        t[3][c] = d1 - d3;
    };
    float *arow = As + ltile * ROW + 4 * quad;
    float *brow = Bs + (tid >> 2) * ROW + 4 * (tid & 3);
    // the A row of position pos (from the row-transformed patch) / the B row of position pos (from staging slot)
    auto store_a = [&](int pos) {
        const int xi = pos >> 2, nu = pos & 3;
        const F4 v = nu == 0 ? t[xi][0] - t[xi][2] : nu == 1 ? t[xi][1] + t[xi][2]
                     : nu == 2 ? t[xi][2] - t[xi][1] : t[xi][1] - t[xi][3];
        st4(arow + pos * MT * ROW, v);
    };
    auto store_b = [&](int pos, int slot) { *reinterpret_cast<float4 *>(brow + pos * NT * ROW) = pb[slot]; };

    f32x16 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    asm volatile("" : "+v"(acc[15]));

    const float *Aw = As + (wm * 32 + (lane & 31)) * ROW + (lane >> 5) * 4;
    const float *Bw = Bs + (wn * 32 + (lane & 31)) * ROW + (lane >> 5) * 4;
    // One group = 8 positions x 2 k-halves = 8 "pair steps": two positions (two independent accumulators) at one k
    // half, 4 fragment reads + 8 MFMAs issued alternately on the two accumulators.  After every MFMA comes a "filler
    // slot" (64 per group) that the caller fills with pieces of the NEXT data's preparation: between MFMAs on
    // DIFFERENT accumulators an extra instruction costs ~6 cycles, between two dependent ones ~43
    // (MI355X_MICROARCH.md), and consecutive MFMAs here always differ.  The fragments of pair step s+1 are read before
    // the MFMAs of pair step s are issued and pinned there (one wave per SIMD: nobody else hides the LDS latency, and
    // left free hipcc sinks the reads to just before their use).
    auto compute = [&](auto group_c, auto &&filler) {
        constexpr int g = decltype(group_c)::value;
        constexpr int PS = 8;  // pair steps: (position pair pp = s >> 1, k half kq = s & 1)
        float4 fa[2][2], fb[2][2];
        auto read_frags = [&](int s, int set) {
            const int q0 = 8 * g + 2 * (s >> 1), kq = s & 1;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                fa[set][u] = *reinterpret_cast<const float4 *>(Aw + (q0 + u) * MT * ROW + kq * 8);
                fb[set][u] = *reinterpret_cast<const float4 *>(Bw + (q0 + u) * NT * ROW + kq * 8);
            }
        };
        read_frags(0, 0);
#pragma unroll
        for (int s = 0; s < PS; ++s) {
            if (s + 1 < PS) read_frags(s + 1, (s + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const int q0 = 8 * g + 2 * (s >> 1), q1 = q0 + 1;
            const float4 a0 = fa[s & 1][0], b0 = fb[s & 1][0], a1 = fa[s & 1][1], b1 = fb[s & 1][1];
            acc[q0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[q0], 0, 0, 0);
            filler(8 * s + 0);
            acc[q1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc[q1], 0, 0, 0);
            filler(8 * s + 1);
            acc[q0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[q0], 0, 0, 0);
            filler(8 * s + 2);
            acc[q1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc[q1], 0, 0, 0);
            filler(8 * s + 3);
            acc[q0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[q0], 0, 0, 0);
            filler(8 * s + 4);
            acc[q1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc[q1], 0, 0, 0);
            filler(8 * s + 5);
            acc[q0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[q0], 0, 0, 0);
            filler(8 * s + 6);
            acc[q1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc[q1], 0, 0, 0);
            filler(8 * s + 7);
        }
    };
    using G0 = std::integral_constant<int, 0>;
    using G1 = std::integral_constant<int, 1>;
    static_assert(KC == 16, "the filler plans below are laid out for 16 channels per K step");

    // prologue: step 0 complete in LDS
#pragma unroll
    for (int i = 0; i < 16; ++i) load_a1(0, true, i);
#pragma unroll
    for (int c = 0; c < 4; ++c) row_transform_col(c);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int j = 0; j < 8; ++j) load_b1(0, 8 * h + j, j);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            store_a(8 * h + j);
            store_b(8 * h + j, j);
        }
    }
    __syncthreads();
    WINO_STAMP();  // 1: prologue done
    // The 256 accumulators fill the accumulator file, so the loop must not give the compiler a reason to copy them (a
    // branch that selects between two multiply blocks does: it then spills accumulators to scratch).  Hence the
    // rotation: first half of step 0 before the loop, one straight-line trip = [second half of step k, first half of
    // step k+1], second half of the last step after it.
    //
    // Filler plans (slot = position in the stream of 64 MFMAs of a half):
    //   second half of step k (multiplies g1(k)):       first half of step k+1 (multiplies g0(k+1)):
    //     0-3   row transform of patch(k+1)               0-7   request filter rows of g1(k+1)
    //     8-15  request filter rows of g0(k+1)            12-19 store A rows of g1(k+1)
    //     16-23 store A rows of g0(k+1)                   40-47 store filter rows of g1(k+1)
    //     24-39 request patch(k+2)
    //     48-55 store filter rows of g0(k+1)
    // vmcnt retires in order, so filter rows are requested BEFORE the patches that follow them in the same half, and
    // the waits in front of their stores leave those 16 patch loads in flight.
    const int nsteps = p.cblocks;
#pragma unroll
    for (int i = 0; i < 16; ++i) load_a1(1, nsteps > 1, i);
    compute(G0{}, [](int) {});
    __syncthreads();
    WINO_STAMP();  // 2: first half of step 0 done
#ifdef WINO_TRACE
    unsigned long long sum_g1 = 0, sum_g0 = 0;
#endif
    for (int k = 0; k + 1 < nsteps; ++k) {
        const bool more2 = k + 2 < nsteps;
#ifdef WINO_TRACE
        const unsigned long long h0 = __builtin_readcyclecounter();
#endif
        compute(G1{}, [&](int slot) {
            if (slot < 4) row_transform_col(slot);
            else if (slot >= 8 && slot < 16) load_b1(k + 1, slot - 8, slot - 8);
            else if (slot >= 16 && slot < 24) store_a(slot - 16);
            else if (slot >= 24 && slot < 40) load_a1(k + 2, more2, slot - 24);
            else if (slot >= 48 && slot < 56) store_b(slot - 48, slot - 48);
        });
        __syncthreads();
        if (k == 0) WINO_STAMP();  // 3: second half of step 0 (with fillers) done
#ifdef WINO_TRACE
        const unsigned long long h1 = __builtin_readcyclecounter();
        sum_g1 += h1 - h0;
#endif
        compute(G0{}, [&](int slot) {
            if (slot < 8) load_b1(k + 1, 8 + slot, slot);
            else if (slot >= 12 && slot < 20) store_a(8 + slot - 12);
            else if (slot >= 40 && slot < 48) store_b(8 + slot - 40, slot - 40);
        });
        __syncthreads();
        if (k == 0) WINO_STAMP();  // 4: first half of step 1 (with fillers) done
#ifdef WINO_TRACE
        sum_g0 += __builtin_readcyclecounter() - h1;
#endif
    }
#ifdef WINO_TRACE
    if (p.trace && threadIdx.x == 0) {
        p.trace[(size_t)(gridDim.x + blockIdx.x) * 8] = sum_g1;
        p.trace[(size_t)(gridDim.x + blockIdx.x) * 8 + 1] = sum_g0;
    }
#endif
    WINO_STAMP();  // 5: loop done
    compute(G1{}, [](int) {});
    // (the 16-pass MFMA needs 18 wait states before its result is read; made explicit as in conv_mfma.hip -- one
    // wait, tied to every accumulator so that no read is scheduled above it)
    asm volatile("s_nop 15\n\ts_nop 7"
                 : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]),
                   "+a"(acc[7]), "+a"(acc[8]), "+a"(acc[9]), "+a"(acc[10]), "+a"(acc[11]), "+a"(acc[12]),
                   "+a"(acc[13]), "+a"(acc[14]), "+v"(acc[15]));

    WINO_STAMP();  // 6: last half done
    // ---- output transform + bias + ReLU + store.  Accumulator element e of a lane: tile row (e&3) + 8(e>>2) + 4(lane>>5)
    // of the wave's 32, output channel n0 + 32 wn + (lane & 31).
    const int n = n0 + wn * 32 + (lane & 31);
    const float bias = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int tt = t0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        float s0[4], s1[4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            s0[nu] = acc[0 * 4 + nu][e] + acc[1 * 4 + nu][e] + acc[2 * 4 + nu][e];
            s1[nu] = acc[1 * 4 + nu][e] - acc[2 * 4 + nu][e] - acc[3 * 4 + nu][e];
        }
        float y00 = s0[0] + s0[1] + s0[2] + bias, y01 = s0[1] - s0[2] - s0[3] + bias;
        float y10 = s1[0] + s1[1] + s1[2] + bias, y11 = s1[1] - s1[2] - s1[3] + bias;
        if (p.relu) {
            y00 = fmaxf(y00, 0.f);
            y01 = fmaxf(y01, 0.f);
            y10 = fmaxf(y10, 0.f);
            y11 = fmaxf(y11, 0.f);
        }
        if (tt < p.T && n < p.N) {
            const WinoTile wt = wino_tile(p, tt);
            float *o = p.y + ((size_t)(wt.img * p.H + wt.a + 2 * p.dil * wt.ty) * p.W + wt.b + 2 * p.dil * wt.tx) * p.N + n;
            o[0] = y00;
            o[(size_t)p.dil * p.N] = y01;
            o[(size_t)p.dil * p.W * p.N] = y10;
            o[(size_t)p.dil * p.W * p.N + (size_t)p.dil * p.N] = y11;
        }
    }
#ifdef WINO_TRACE
    WINO_STAMP();  // 7: epilogue issued
    if (p.trace && tid == 0)
        for (int i = 0; i < 8; ++i) p.trace[(size_t)blockIdx.x * 8 + i] = i < nts ? ts[i] : 0;
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// Eight-wave variant: the same workgroup tile, LDS layout and K-step schedule, but 512 threads -- two waves per SIMD.
// One wave per SIMD means every LDS / memory instruction of the fill costs the wave its issue time (tools/
// mfma_peak.py --mix: 4 ds_read_b128 per MFMA run at 75 TFLOP/s with one wave per SIMD and 124 with four); with a
// second wave on the SIMD those issue slots overlap the other wave's MFMAs.  To fit two waves into the register file
// a wave keeps HALF the positions (4 of the 8 of each group: 128 accumulator registers); the fill is split the same
// way -- a thread loads / transforms / stores 2 of its slot's 4 channels and 4 of a group's 8 filter rows -- and the
// output transform, being linear, is done as two partial sums that the wave pair exchanges through LDS once.
template <bool SPLIT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino_conv8_kernel(const WinoParams p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;                   // [16][MT][ROW]
    float *Bs = lds + 16 * MT * ROW;   // [16][NT][ROW]
    const int xcd = blockIdx.x & 7, l_ = blockIdx.x >> 3;
    const int nb = l_ % p.nblocks;
    int l2_ = l_ / p.nblocks, slice = 0;
    if constexpr (SPLIT) {
        slice = l2_ % p.nslices;
        l2_ /= p.nslices;
    }
    const int mb = l2_ * 8 + xcd;
    if (mb >= p.mblocks) return;  // block-uniform
    // (a slice's channel steps are reached by shifting the operands' byte offsets: the loop below counts from 0)
    const unsigned a_slice = SPLIT ? (unsigned)(slice * p.steps) * KC * 4u : 0u;
    const int n0 = nb * NT, t0 = mb * MT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 2 x 2 wave grid, position half: wave-uniform by construction -- said so, or every load whose scalar offset
    // depends on them becomes a waterfall loop
    const int wq = __builtin_amdgcn_readfirstlane(wave & 3), ph = __builtin_amdgcn_readfirstlane(wave >> 2);
    const int wm = wq >> 1, wn = wq & 1;
    const int slot_id = tid & 255;            // loader slot (tile or filter row, channel quad); ph doubles as its half
    const __amdgpu_buffer_rsrc_t ru =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, (int)p.ubytes, 0x00020000);

    // ---- A loader: thread = (tile, channel quad, channel pair of the quad)
    const int ltile = slot_id >> 2, quad = slot_id & 3;
    unsigned aoff[16];
    {
        const int t = t0 + ltile;
        const WinoTile wt = wino_tile(p, t < p.T ? t : 0);
        const int y0 = 2 * wt.ty - 1, x0 = 2 * wt.tx - 1;  // sub-grid coordinates of the patch's corner
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) {
                const bool ok = t < p.T && y0 + r >= 0 && y0 + r < 2 * p.th && x0 + sx >= 0 && x0 + sx < 2 * p.tw;
                const int yy = wt.a + p.dil * (y0 + r), xx = wt.b + p.dil * (x0 + sx);
                aoff[r * 4 + sx] = ok ? ((unsigned)((wt.img * p.H + yy) * p.W + xx) * (unsigned)p.C + 4u * quad + 2u * ph) * 4u + a_slice
                                      : p.xbytes;
            }
    }
    // ---- B loader: thread = (filter row n, channel quad); it handles positions 4 ph .. 4 ph + 3 of a group
    const unsigned bstride = (unsigned)p.N * KC * 4u;
    const unsigned boff = n0 + (slot_id >> 2) < p.N
        ? (unsigned)((n0 + (slot_id >> 2)) * KC + 4 * (slot_id & 3)) * 4u + 4u * (unsigned)ph * bstride +
              (SPLIT ? (unsigned)(slice * p.steps) * 16u * bstride : 0u)
        : p.ubytes;

    f32x2 pa[16];
    float4 pb[4];
    // (a request past the last channel block goes through a zero-length descriptor: a scalar select, where a select
    // on the offset would be one vector instruction per request)
    auto load_a1 = [&](int cb, bool live, int i) {
        const __amdgpu_buffer_rsrc_t r =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, live ? (int)p.xbytes : 0, 0x00020000);
        pa[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, aoff[i], live ? (unsigned)cb * KC * 4u : 0u, 0));
    };
    auto load_b1 = [&](int cb, int pos, int slot) {
        pb[slot] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  ru, boff, ((unsigned)cb * 16u + (unsigned)pos) * bstride, 0));
    };
    // (the packed adds are spelled out: hipcc scalarises about half of them otherwise, and with two waves per SIMD a
    // vector instruction is matrix-pipe time)
    auto padd = [](f32x2 a, f32x2 b) {
        f32x2 r;
        asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    };
    auto psub = [](f32x2 a, f32x2 b) {
        f32x2 r;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
        return r;
    };
    f32x2 t[4][4];
    auto row_transform_col = [&](int c) {
        const f32x2 d0 = pa[c], d1 = pa[4 + c], d2 = pa[8 + c], d3 = pa[12 + c];
        t[0][c] = psub(d0, d2);
        t[1][c] = padd(d1, d2);
        t[2][c] = psub(d2, d1);
        t[3][c] = psub(d1, d3);
    };
    float *arow = As + ltile * ROW + 4 * quad + 2 * ph;
    float *brow = Bs + (slot_id >> 2) * ROW + 4 * (slot_id & 3) + 4 * ph * NT * ROW;  // + this thread's position half
    auto store_a = [&](int pos) {
        const int xi = pos >> 2, nu = pos & 3;
        const f32x2 v = nu == 0 ? psub(t[xi][0], t[xi][2]) : nu == 1 ? padd(t[xi][1], t[xi][2])
                        : nu == 2 ? psub(t[xi][2], t[xi][1]) : psub(t[xi][1], t[xi][3]);
        *reinterpret_cast<f32x2 *>(arow + pos * MT * ROW) = v;
    };
    auto store_b = [&](int pos, int slot) { *reinterpret_cast<float4 *>(brow + pos * NT * ROW) = pb[slot]; };

    // accumulators: position 8 g + 4 ph + r  ->  acc[4 g + r]
    f32x16 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;

    // (the wave's position half is folded into the base addresses: every fragment offset below is a literal)
    const float *Aw = As + (wm * 32 + (lane & 31)) * ROW + (lane >> 5) * 4 + 4 * ph * MT * ROW;
    const float *Bw = Bs + (wn * 32 + (lane & 31)) * ROW + (lane >> 5) * 4 + 4 * ph * NT * ROW;
    // one group for this wave = 4 positions x 2 k-halves = 4 pair steps of 8 MFMAs, a filler slot after every MFMA
    auto compute = [&](auto group_c, auto &&filler) {
        constexpr int g = decltype(group_c)::value;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r0 = 2 * (s >> 1), kq = s & 1;
            const int q0 = 8 * g + r0;  // LDS position of the pair, relative to the wave's half
            // (measured: reading the next pair step's fragments ahead, or letting hipcc move these reads, both cost 4 %:
            // the other wave of the SIMD covers the latency, extra live registers and reordered waits do not pay)
            __builtin_amdgcn_sched_barrier(0);
            float4 a0 = *reinterpret_cast<const float4 *>(Aw + q0 * MT * ROW + kq * 8);
            float4 b0 = *reinterpret_cast<const float4 *>(Bw + q0 * NT * ROW + kq * 8);
            float4 a1 = *reinterpret_cast<const float4 *>(Aw + (q0 + 1) * MT * ROW + kq * 8);
            float4 b1 = *reinterpret_cast<const float4 *>(Bw + (q0 + 1) * NT * ROW + kq * 8);
            f32x16 &c0 = acc[4 * g + r0], &c1 = acc[4 * g + r0 + 1];
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, c0, 0, 0, 0);
            filler(8 * s + 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, c1, 0, 0, 0);
            filler(8 * s + 1);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, c0, 0, 0, 0);
            filler(8 * s + 2);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, c1, 0, 0, 0);
            filler(8 * s + 3);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, c0, 0, 0, 0);
            filler(8 * s + 4);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, c1, 0, 0, 0);
            filler(8 * s + 5);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, c0, 0, 0, 0);
            filler(8 * s + 6);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, c1, 0, 0, 0);
            filler(8 * s + 7);
        }
    };
    using G0 = std::integral_constant<int, 0>;
    using G1 = std::integral_constant<int, 1>;

    // prologue: step 0 complete in LDS
#pragma unroll
    for (int i = 0; i < 16; ++i) load_a1(0, true, i);
#pragma unroll
    for (int c = 0; c < 4; ++c) row_transform_col(c);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int j = 0; j < 4; ++j) load_b1(0, 8 * h + j, j);
#pragma unroll
        for (int j = 0; j < 8; ++j) store_a(8 * h + j);
#pragma unroll
        for (int j = 0; j < 4; ++j) store_b(8 * h + j, j);
    }
    __syncthreads();
    // Rotated loop as in the four-wave kernel.  Filler plans (32 slots per half):
    //   second half of step k (multiplies g1(k)):        first half of step k+1 (multiplies g0(k+1)):
    //     4-7   row transform of patch(k+1)                 0-3   request this thread's filter rows of g1(k+1)
    //     8-15  store A rows of g0(k+1)                     6-13  store A rows of g1(k+1)
    //     16-19 store filter rows of g0(k+1)                14-19 request patch(k+2), pixels 8-13
    //     20-27 request patch(k+2), pixels 0-7              20-23 store filter rows of g1(k+1)
    //                                                       24-25 request patch(k+2), pixels 14-15
    //                                                       26-29 request filter rows of g0(k+2)
    // (measured: spreading the requests over both halves is worth 4 % of the kernel; vmcnt retires in order, so a
    // store is never placed behind younger requests than the ones it needs)
    const int nsteps = SPLIT ? p.steps : p.cblocks;
#pragma unroll
    for (int i = 0; i < 16; ++i) load_a1(1, nsteps > 1, i);
    compute(G0{}, [&](int slot) {
        if (slot >= 26 && slot < 30) load_b1(1, slot - 26, slot - 26);  // g0's filter rows of step 1 (zeros past the end)
    });
    __syncthreads();
    for (int k = 0; k + 1 < nsteps; ++k) {
        const bool more2 = k + 2 < nsteps;
        compute(G1{}, [&](int slot) {
            if (slot >= 4 && slot < 8) row_transform_col(slot - 4);
            else if (slot >= 8 && slot < 16) store_a(slot - 8);
            else if (slot >= 16 && slot < 20) store_b(slot - 16, slot - 16);
            else if (slot >= 20 && slot < 28) load_a1(k + 2, more2, slot - 20);  // pixels 0-7
        });
        __syncthreads();
        compute(G0{}, [&](int slot) {
            if (slot < 4) load_b1(k + 1, 8 + slot, slot);
            else if (slot >= 6 && slot < 14) store_a(8 + slot - 6);
            else if (slot >= 14 && slot < 20) load_a1(k + 2, more2, 8 + slot - 14);  // pixels 8-13
            else if (slot >= 20 && slot < 24) store_b(8 + slot - 20, slot - 20);
            else if (slot >= 24 && slot < 26) load_a1(k + 2, more2, 14 + slot - 24);  // pixels 14-15
            else if (slot >= 26 && slot < 30) load_b1(k + 2, slot - 26, slot - 26);  // g0's rows of the step after
        });
        __syncthreads();
    }
    compute(G1{}, [](int) {});
    asm volatile("s_nop 15\n\ts_nop 7"
                 : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]),
                   "+a"(acc[7]));

    // ---- output transform, split over the wave pair.  This wave holds the transformed-domain rows xi = ph and ph + 2
    // (positions 4 xi + nu): acc[nu] = M[ph][nu], acc[4 + nu] = M[ph + 2][nu].  With s0 = M0 + M1 + M2 and
    // s1 = M1 - M2 - M3 per column nu, the wave's partial sums are
    //     ph 0 (rows 0, 2):  s0' = M0 + M2,  s1' = -M2          ph 1 (rows 1, 3):  s0' = M1,  s1' = M1 - M3
    // Output row 0 (y00, y01 from s0) is finished by the ph = 0 wave, row 1 (y10, y11 from s1) by the ph = 1 wave; each
    // hands the other its partial of the row it does not finish (2 values per element) through LDS.
    __syncthreads();  // every wave is done reading the A / B tiles
    float *xch = lds + (size_t)wq * (2 * 2 * 16 * 64);  // [target ph][value][element][lane]
    const int n = n0 + wn * 32 + (lane & 31);
    const float bias = (!SPLIT && p.bias && n < p.N) ? p.bias[n] : 0.f;  // (SPLIT: partial sums; bias / ReLU in the finish)
    float *ybase = SPLIT ? p.part + (size_t)slice * p.yfloats : p.y;
    float mine[16][2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float s0[4], s1[4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            const float ma = acc[nu][e], mb2 = acc[4 + nu][e];  // rows ph and ph + 2
            if (ph == 0) {
                s0[nu] = ma + mb2;
                s1[nu] = -mb2;
            } else {
                s0[nu] = ma;
                s1[nu] = ma - mb2;
            }
        }
        const float y00 = s0[0] + s0[1] + s0[2], y01 = s0[1] - s0[2] - s0[3];
        const float y10 = s1[0] + s1[1] + s1[2], y11 = s1[1] - s1[2] - s1[3];
        // keep the row this wave finishes, hand over the other one
        mine[e][0] = ph == 0 ? y00 : y10;
        mine[e][1] = ph == 0 ? y01 : y11;
        float *dst = xch + (size_t)((1 - ph) * 2) * 16 * 64 + e * 64 + lane;
        dst[0] = ph == 0 ? y10 : y00;
        dst[16 * 64] = ph == 0 ? y11 : y01;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int tt = t0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const float *src = xch + (size_t)(ph * 2) * 16 * 64 + e * 64 + lane;
        float ya = mine[e][0] + src[0] + bias, yb = mine[e][1] + src[16 * 64] + bias;
        if (!SPLIT && p.relu) {
            ya = fmaxf(ya, 0.f);
            yb = fmaxf(yb, 0.f);
        }
        if (tt < p.T && n < p.N) {
            const WinoTile wt = wino_tile(p, tt);
            float *o = ybase + ((size_t)(wt.img * p.H + wt.a + p.dil * (2 * wt.ty + ph)) * p.W + wt.b + 2 * p.dil * wt.tx) * p.N + n;
            o[0] = ya;
            o[(size_t)p.dil * p.N] = yb;
        }
    }
}

}  // namespace

static unsigned long long *g_wino_trace = nullptr;
extern "C" void mpsr_debug_set_wino_trace(void *buf) { g_wino_trace = static_cast<unsigned long long *>(buf); }
namespace mpsr { extern std::atomic<int> g_wino_waves; }
extern "C" void mpsr_debug_set_wino_waves(int waves) { mpsr::g_wino_waves = waves; }

namespace mpsr {

// floats of scratch conv3x3_winograd needs behind `ws` (the transformed filters)
size_t winograd_scratch_floats(int C, int N) { return (size_t)16 * N * C; }

bool winograd_applies(int H, int W, int C, int N) { return H % 2 == 0 && W % 2 == 0 && C % KC == 0 && C >= KC && N >= 1; }
// at dilation d: the d^2 pixel sub-grids of H / d x W / d pixels must be whole and even-sized
bool winograd_applies_dilated(int H, int W, int C, int N, int dilation)
{
    return dilation >= 1 && H % (2 * dilation) == 0 && W % (2 * dilation) == 0 && C % KC == 0 && C >= KC && N >= 1;
}

std::atomic<int> g_wino_waves{8};  // 4: the one-wave-per-SIMD kernel, 8: the two-waves-per-SIMD variant

// winograd3z.hip: K slices of a small launch and the launch that adds them
int winograd_slices(long long blocks, int want_blocks, int cblocks, int min_steps, size_t part_floats, size_t y_floats);
int winograd_finish_slices(const float *part, const float *bias, float *y, size_t y_floats, int nslices, int N, int relu,
                           hipStream_t s);

int conv3x3_winograd(const float *x, int B, int H, int W, int C, const float *w, const float *bias, int relu, float *y,
                     int N, float *ws, size_t ws_floats, hipStream_t s, int dilation)
{
    MPSR_REQUIRE(winograd_applies_dilated(H, W, C, N, dilation),
                 "conv3x3_winograd: needs H, W multiples of 2 x dilation and C %% %d == 0", KC);
    if (ws_floats < winograd_scratch_floats(C, N) || !ws)
        return fail(MPSR_ERR_WORKSPACE, "conv3x3_winograd: scratch holds %zu floats, needs %zu", ws_floats,
                    winograd_scratch_floats(C, N));
    // (per call: the attribute belongs to the current device; cheap and idempotent)
    MPSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(g_wino_waves.load() == 8
                                                                          ? reinterpret_cast<const void *>(wino_conv8_kernel<false>)
                                                                          : reinterpret_cast<const void *>(wino_conv_kernel)),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsFloats * sizeof(float))));
    // the caller's filter cache (mpsr_net_opts), if the network entry point offered a slot for this layer: the
    // full-image trunk's atrous layers run here (F(2x2,3x3) on their pixel sub-grids), 27 filter transforms per call
    float *u = ws;
    bool ready = false;
    if (g_filter_cache_slot.w == w && g_filter_cache_slot.u && g_filter_cache_slot.floats >= winograd_scratch_floats(C, N)) {
        u = g_filter_cache_slot.u;
        ready = g_filter_cache_slot.holds(FILTER_FORM_WINO2);
    }
    g_filter_cache_slot = FilterCacheSlot();
    if (!ready) {
        const long long total = (long long)N * C;
        hipLaunchKernelGGL(wino_filter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, N, C, u);
        MPSR_CHECK_LAUNCH("wino_filter_kernel");
    }
    WinoParams p;
    p.x = x; p.u = u; p.bias = bias; p.y = y;
    p.B = B; p.H = H; p.W = W; p.C = C; p.N = N;
    p.dil = dilation;
    p.th = H / (2 * dilation); p.tw = W / (2 * dilation);
    p.T = B * dilation * dilation * p.th * p.tw;
    p.cblocks = C / KC;
    p.nblocks = ceil_div(N, NT);
    p.mblocks = ceil_div(p.T, MT);
    p.relu = relu;
    p.xbytes = (unsigned)((long long)B * H * W * C * 4);
    p.ubytes = (unsigned)(winograd_scratch_floats(C, N) * 4);
    p.trace = g_wino_trace;
    const long long blocks = 8LL * ceil_div(p.mblocks, 8) * p.nblocks;
    if (blocks > 0x7fffffffLL) return fail(MPSR_ERR_UNSUPPORTED, "conv3x3_winograd: grid too large");
    // launches too small to fill the chip (the full-image trunk's atrous layers at ONE image: 96 workgroups) are cut
    // along K like the sixteen-product kernel's (winograd3z.hip): partial outputs behind the transformed filters in `ws`
    p.part = nullptr; p.nslices = 1; p.steps = p.cblocks;
    p.yfloats = (size_t)B * H * W * N;
    if (g_wino_waves.load() == 8) {
        float *part = ws;
        size_t part_floats = ws_floats;
        if (u == ws) {
            const size_t off = align_up(winograd_scratch_floats(C, N), 64);
            part = ws + off;
            part_floats = ws_floats > off ? ws_floats - off : 0;
        }
        const bool splittable = N % 4 == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)part & 15) == 0 &&
                                (!bias || ((uintptr_t)bias & 15) == 0);
        const int slices = winograd_slices(blocks, 256, p.cblocks, 2, splittable ? part_floats : 0, p.yfloats);
        if (slices > 1) {
            p.part = part; p.nslices = slices; p.steps = p.cblocks / slices;
            MPSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wino_conv8_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsFloats * sizeof(float))));
            hipLaunchKernelGGL(wino_conv8_kernel<true>, dim3((unsigned)(blocks * slices)), dim3(512),
                               kLdsFloats * sizeof(float), s, p);
            MPSR_CHECK_LAUNCH("wino_conv8_kernel");
            return winograd_finish_slices(part, bias, y, p.yfloats, slices, N, relu, s);
        }
        hipLaunchKernelGGL(wino_conv8_kernel<false>, dim3((unsigned)blocks), dim3(512), kLdsFloats * sizeof(float), s, p);
    } else
        hipLaunchKernelGGL(wino_conv_kernel, dim3((unsigned)blocks), dim3(256), kLdsFloats * sizeof(float), s, p);
    MPSR_CHECK_LAUNCH("wino_conv_kernel");
    return MPSR_OK;
}

}  // namespace mpsr
