// Atrous 3x3 convolution whose pixel sub-grids are 3x3 (H = W = 3 * dilation: ResNet-101 block3's conv2 at output
// stride 4 -- 12x12 maps, dilation 4, 23 launches per step; reference graph object_detection/nets/resnet_v1.py:116-127,
// resnet_utils.py:194-196) by Winograd F(3x3, 3x3) on fp32 MFMA.
//
// At dilation d the pixels (d i + a, d j + b), i, j = 0..2, of one (a, b) form an independent 3x3 image: its nine
// outputs only read its own nine inputs, everything a tap reaches outside is SAME-padding zeros.  So every sub-grid is
// exactly ONE F(3x3,3x3) tile whose 5x5 input patch has an all-zero ring:
//     Y (3x3) = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A,        d = the 3x3 sub-grid inside a ring of zeros.
// 25 products per (channel pair, sub-grid) where the direct form has 81 and the border-class implicit GEMM
// (conv_mfma.hip), which already skips the out-of-image taps, executes 49: the matrix pipes issue 0.51 of what they do
// today.  No halo, no overlap between tiles, 9 loads per patch.  Points 0, 1, -1, 2, inf; rows of B^T / G scaled by
// (2, 2, 6, 6, 1) so that the input transform has integer constants (7 operations per 3 -> 5 transform).  Error
// against float64 ~5e-6 of the tensor scale.
//
// Kernel structure = winograd4.hip's (see there): workgroup = 32 tiles x 64 output channels x all 25 positions, 512
// threads; K step = 8 channels; A (transformed patches) double buffered in LDS with the 16-byte-half XOR swizzle; B
// fragments straight from the transformed filters (L2) into registers one unit ahead; producers alternate request /
// transform steps; every MFMA followed by its slot of producer work, order pinned.  25 positions over 8 waves: wave
// (pg, nh) owns positions 6 pg .. 6 pg + 5 (pg = 3: .. + 6) x 32 of the 64 channels -- 96 or 112 accumulator
// registers; the two seven-position waves sit on different SIMDs.
#include <atomic>
#include <type_traits>

#include "common.h"
#include "wino3_filter.h"
#include "wino3_transforms.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace f3 {
constexpr int MT = 32, NT = 64, KC = 8, NP = 25;
constexpr int APOS = MT * KC;           // floats per position of an A buffer (256)
constexpr int ABUF = NP * APOS;         // one A buffer (6400 floats = 25 KB)
constexpr int MEXF = NP * 16 * NT;      // epilogue exchange of one round: 25 x 16 tiles x 64 channels (100 KB)
constexpr int LDSF = MEXF > 2 * ABUF ? MEXF : 2 * ABUF;
constexpr unsigned OOB = 0x80000000u;
}  // namespace f3

struct Wino3Params {
    const float *x, *u, *bias;
    float *y;
    int B, H, W, C, N, dil, T;  // T = B * dil * dil * th * th tiles
    int th;                     // 3x3 tiles per sub-grid side: 1 = the sub-grid IS one tile (zero ring), > 1 = tiles with halos
    int cblocks, nblocks, mblocks, relu;
    unsigned xbytes, ubytes;
    // training (data-gradient launches): a tensor shaped like y; an output element is kept where mask[...] > 0 and written
    // as zero elsewhere -- the ReLU gradient of the layer this gradient belongs to, in the epilogue instead of by an
    // elementwise pass over the result
    const float *mask;
};

__global__ __launch_bounds__(256) void wino3_filter_kernel(const float *__restrict__ w, int N, int C, float *__restrict__ u)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < (long long)N * C) mpsr::wino3_filter_one(w, N, C, u, i);
}

using mpsr::w3t::bt5;
using mpsr::w3t::at3;

template <int V>
using IC3 = std::integral_constant<int, V>;

// NPOS = positions of this wave (6 or 7), DG = producer group (0: patches of the even K steps, 1: odd).
// HALO: the pixel sub-grids are larger than one tile (side 3 th, th > 1: ResNet-101 block2's atrous layers, 12x12 at
// dilation 2 -> 6x6 sub-grids of 2x2 tiles): a tile's 5x5 patch then holds real neighbours -- 25 requests per patch
// (those outside the sub-grid through an out-of-range offset) and the full transform, B^T d = the zero-ring form on
// d1..d3 plus 2 d0 in row 0 and d4 in row 4 (9 operations per 1-D pass instead of 7).
template <int NPOS, int DG, bool HALO>
__device__ __forceinline__ void wino3_body(const Wino3Params &p, const int n0, const int t0, float *lds)
{
    using namespace f3;
    constexpr int NU = NPOS == 7 ? 3 : 2;  // units: 3 + 3 (+ 1) positions
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave >> 1, nh = wave & 1;
    const int gpos = 6 * pg;  // first position of the wave
    const int nsteps = p.cblocks;
    const int d = p.dil;

    // ---- A producer: thread = (tile = sub-grid (img, a, b), channel of the step); nine 4-byte requests per patch
    const int lt = 8 * (wave & 3) + (lane >> 3), ch = lane & 7;
    const int th = HALO ? p.th : 1, tpi = d * d * th * th;  // tiles per image
    unsigned abase;
    unsigned voffc[3][3];  // HALO: the thread's offset by (row class, column class) of a patch element: first / inner / last
    // HALO: descriptor moved back by one sub-grid row + column, so that patch element (0, 0) has a non-negative offset
    const unsigned shift = HALO ? (unsigned)(d * p.W + d) * (unsigned)p.C * 4u : 0u;
    char *xback = const_cast<char *>(reinterpret_cast<const char *>(p.x)) - shift;
    {
        const int t = t0 + lt;
        const int img = t / tpi, rem = t - img * tpi;
        const int sub = rem / (th * th), tt = rem - sub * th * th;
        const int a = sub / d, b = sub - a * d, ty = tt / th, tx = tt - ty * th;
        // (+ shift: the offset of patch element (1, 1) = sub-grid pixel (3 ty, 3 tx) from the moved-back base is that of (0, 0))
        abase = t < p.T ? (unsigned)(((img * p.H + a + 3 * d * ty) * p.W + b + 3 * d * tx) * p.C + ch) * 4u : OOB;
        const bool rowc[3] = {ty > 0, true, ty < th - 1}, colc[3] = {tx > 0, true, tx < th - 1};
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) voffc[u][v] = (t < p.T && rowc[u] && colc[v]) ? abase : OOB;
    }
    float pa[25];  // 5x5: the 3x3 data arrives in the middle, the transforms expand it in place (HALO: all 25 arrive)
    auto load_patch1 = [&](int step, int L) __attribute__((always_inline)) {
#ifdef W3_SKIP_LOAD
        if (step > 1) return;
#endif
        const bool live = step < nsteps;
        if constexpr (HALO) {
            const int s_ = L / 5, r_ = L % 5;  // column by column
            const __amdgpu_buffer_rsrc_t rr =
                __builtin_amdgcn_make_buffer_rsrc(xback, 0, live ? (int)(p.xbytes + shift) : 0, 0x00020000);
            const unsigned so = (unsigned)((d * r_ * p.W + d * s_) * p.C + (live ? step : 0) * KC) * 4u;
            pa[5 * r_ + s_] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rr, voffc[r_ == 0 ? 0 : r_ == 4 ? 2 : 1][s_ == 0 ? 0 : s_ == 4 ? 2 : 1],
                                                            so, 0));
        } else {
            const int j = L / 3, i = L % 3;  // column by column
            const __amdgpu_buffer_rsrc_t rr =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, live ? (int)p.xbytes : 0, 0x00020000);
            const unsigned so = (unsigned)((d * i * p.W + d * j) * p.C + (live ? step : 0) * KC) * 4u;
            pa[5 * (i + 1) + (j + 1)] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, abase, so, 0));
        }
    };
    auto vertical = [&](int j) __attribute__((always_inline)) {  // data column j (0..2; HALO: patch column j - 1 + 1 = 0..4) -> all five rows of it
#ifdef W3_SKIP_VALU
        return;
#endif
        float t0_, t1_, t2_, t3_, t4_;
        if constexpr (HALO) {  // j = patch column 0..4
            bt5(pa[5 + j], pa[10 + j], pa[15 + j], t0_, t1_, t2_, t3_, t4_);
            pa[j] = fmaf(2.f, pa[j], t0_);
            pa[5 + j] = t1_;
            pa[10 + j] = t2_;
            pa[15 + j] = t3_;
            pa[20 + j] = pa[20 + j] + t4_;
            return;
        }
        bt5(pa[5 + j + 1], pa[10 + j + 1], pa[15 + j + 1], t0_, t1_, t2_, t3_, t4_);
        pa[j + 1] = t0_;
        pa[5 + j + 1] = t1_;
        pa[10 + j + 1] = t2_;
        pa[15 + j + 1] = t3_;
        pa[20 + j + 1] = t4_;
    };
    auto horizontal = [&](int r) __attribute__((always_inline)) {  // row r (0..4): three values -> five (HALO: five -> five)
#ifdef W3_SKIP_VALU
        return;
#endif
        float t0_, t1_, t2_, t3_, t4_;
        bt5(pa[5 * r + 1], pa[5 * r + 2], pa[5 * r + 3], t0_, t1_, t2_, t3_, t4_);
        if constexpr (HALO) {
            t0_ = fmaf(2.f, pa[5 * r], t0_);
            t4_ = pa[5 * r + 4] + t4_;
        }
        pa[5 * r] = t0_;
        pa[5 * r + 1] = t1_;
        pa[5 * r + 2] = t2_;
        pa[5 * r + 3] = t3_;
        pa[5 * r + 4] = t4_;
    };
    // A[buf][pos][tile][8 channels], 16-byte halves swapped on odd 8-row blocks
    float *awr = lds + lt * 8 + 4 * ((ch >> 2) ^ ((lt >> 3) & 1)) + (ch & 3);
    auto store_a = [&](int buf, int pos) __attribute__((always_inline)) {
#ifndef W3_SKIP_STORE  // (timing experiments only)
        awr[buf * ABUF + pos * APOS] = pa[pos];
#else
        asm volatile("" ::"v"(pa[pos]));
#endif
    };

    // ---- B fragments from the transformed filters, lane = (n = lane & 31, k half = lane >> 5)
    const unsigned bvoff = n0 + 32 * nh + (lane & 31) < p.N
                               ? (unsigned)((n0 + 32 * nh + (lane & 31)) * KC + 4 * (lane >> 5)) * 4u : OOB;
    const unsigned bpstride = (unsigned)p.N * KC * 4u;
    float4 fb[NU][3];
    auto load_b1 = [&](int step, int q, int set, int j) __attribute__((always_inline)) {  // position q of the wave
#ifdef W3_SKIP_BLOAD
        if (step > 0) return;
#endif
        const bool live = step < nsteps;
        const __amdgpu_buffer_rsrc_t rr =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, live ? (int)p.ubytes : 0, 0x00020000);
        const unsigned so = ((unsigned)(live ? step : 0) * (unsigned)NP + (unsigned)(gpos + q)) * bpstride;
        fb[set][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, bvoff, so, 0));
    };
    const float *ard = lds + gpos * APOS + (lane & 31) * 8 + 4 * ((lane >> 5) ^ (((lane & 31) >> 3) & 1));
    float4 fa[3];

    f32x16 acc[NPOS];
#pragma unroll
    for (int q = 0; q < NPOS; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;

    // One K step: units of 3, 3 (and 1) positions; within a unit k = 0..3 of the fragments outermost, the unit's
    // positions round-robin.  MFMA number m of the step is followed by duty(m) for m < 24.  The B fragments of the next
    // unit are requested in the first slots of a unit (register set = unit index).
    auto kstep = [&](int s, auto buf_c, auto &&duty) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_c)::value;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int np = u < 2 ? 3 : 1;  // positions of this unit
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < np) fa[j] = *reinterpret_cast<const float4 *>(ard + buf * ABUF + (3 * u + j) * APOS);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (j >= np) continue;
                    const int m = 12 * u + np * k + j;
                    const float av = k == 0 ? fa[j].x : k == 1 ? fa[j].y : k == 2 ? fa[j].z : fa[j].w;
                    const float bv = k == 0 ? fb[u][j].x : k == 1 ? fb[u][j].y : k == 2 ? fb[u][j].z : fb[u][j].w;
                    acc[3 * u + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[3 * u + j], 0, 0, 0);
                    if (k == 0) {  // request the next unit's fragment j (of this step, or unit 0 of the next step)
                        const int un = (u + 1) % NU, npn = un < 2 ? 3 : 1;
                        if (j < npn) load_b1(u + 1 < NU ? s : s + 1, 3 * un + j, un, j);
                        if (np == 1 && npn == 3) {  // a one-position unit requests all three of the next
                            load_b1(u + 1 < NU ? s : s + 1, 3 * un + 1, un, 1);
                            load_b1(u + 1 < NU ? s : s + 1, 3 * un + 2, un, 2);
                        }
                    }
                    if (m < 24) duty(m);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    };
    // producer duties: request = the nine loads in slots 3..11; transform = slots 0..2 the three column transforms,
    // 3..7 the five row transforms, a row's five stores in the two slots behind it... (row r: slots 4 + r and 5 + r
    // would collide with the next row's arithmetic only in issue order, not in data: stores of row r use finished values)
    auto request = [&](int step) __attribute__((always_inline)) {
        return [&, step](int slot) __attribute__((always_inline)) {
            if constexpr (HALO) {  // 25 requests: one per slot, the last slot two
                load_patch1(step, slot);
                if (slot == 23) load_patch1(step, 24);
            } else {
                if (slot >= 3 && slot < 12) load_patch1(step, slot - 3);
            }
        };
    };
    auto transform = [&](auto buf_c) __attribute__((always_inline)) {
        return [&](int slot) __attribute__((always_inline)) {
            constexpr int buf = decltype(buf_c)::value;
            if constexpr (HALO) {  // slots 0..4 the five column transforms, 5, 7, .. 13 the rows, a row's stores behind it
                if (slot < 5) vertical(slot);
                else if (slot < 15 && ((slot - 5) & 1) == 0) horizontal((slot - 5) >> 1);
                if (slot >= 6 && slot < 16) {
                    const int r = (slot - 6) >> 1;
                    if (((slot - 6) & 1) == 0) {
                        store_a(buf, 5 * r);
                        store_a(buf, 5 * r + 1);
                        store_a(buf, 5 * r + 2);
                    } else {
                        store_a(buf, 5 * r + 3);
                        store_a(buf, 5 * r + 4);
                    }
                }
                return;
            }
            if (slot < 3) vertical(slot);
            else if (slot < 13 && ((slot - 3) & 1) == 0) horizontal((slot - 3) >> 1);  // slots 3, 5, 7, 9, 11
            // row r is complete after slot 3 + 2 r: its five stores go to slots 4 + 2 r (three) and 5 + 2 r... the
            // last row's second half to slot 13
            if (slot >= 4 && slot < 14) {
                const int r = (slot - 4) >> 1;
                if (((slot - 4) & 1) == 0) {
                    store_a(buf, 5 * r);
                    store_a(buf, 5 * r + 1);
                    store_a(buf, 5 * r + 2);
                } else {
                    store_a(buf, 5 * r + 3);
                    store_a(buf, 5 * r + 4);
                }
            }
        };
    };

    // ---- prologue: B of (step 0, unit 0); A of step 0 by waves 0-3; the patch of step 1 requested by waves 4-7
#pragma unroll
    for (int j = 0; j < 3; ++j) load_b1(0, j, 0, j);
#pragma unroll
    for (int i = 0; i < (HALO ? 25 : 9); ++i) load_patch1(DG, i);
    if (DG == 0) {
#pragma unroll
        for (int j = 0; j < (HALO ? 5 : 3); ++j) vertical(j);
#pragma unroll
        for (int r = 0; r < 5; ++r) horizontal(r);
#pragma unroll
        for (int i = 0; i < 25; ++i) store_a(0, i);
    }
    __syncthreads();

    // ---- K loop, two steps per trip (roles as in winograd4.hip)
#ifdef W3_NO_BARRIER  // (timing experiment only: wrong results)
#define W3_SYNC() do { } while (0)
#else
#define W3_SYNC() __syncthreads()
#endif
    if (DG == 0) {
        for (int s = 0; s < nsteps; s += 2) {
            kstep(s, IC3<0>{}, request(s + 2));
            W3_SYNC();
            kstep(s + 1, IC3<1>{}, transform(IC3<0>{}));
            W3_SYNC();
        }
    } else {
        for (int s = 0; s < nsteps; s += 2) {
            kstep(s, IC3<0>{}, transform(IC3<1>{}));
            W3_SYNC();
            kstep(s + 1, IC3<1>{}, request(s + 3));
            W3_SYNC();
        }
    }
    int tid2 = tid;
    if constexpr (NPOS == 7)
        asm volatile("s_nop 15\n\ts_nop 7"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                       "+v"(tid2));
    else
        asm volatile("s_nop 15\n\ts_nop 7"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(tid2));

    // ---- epilogue: two rounds of 16 tiles.  M[pos][tile][64 n] through LDS; thread (tile = tid >> 5, n = tid & 31 and
    // + 32) gathers the 25 positions of its two (tile, n) pairs, A^T M A, bias, ReLU, nine strided pixels out.
    const int lane2 = tid2 & 63;
    float *mwr = lds + gpos * (16 * NT) + (4 * (lane2 >> 5)) * NT + 32 * nh + (lane2 & 31);
    const float *mrd = lds + (tid2 >> 5) * NT + (tid2 & 31);
    float bias2[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = n0 + (tid2 & 31) + 32 * h;
        bias2[h] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (round) __syncthreads();
#pragma unroll
        for (int q = 0; q < NPOS; ++q)
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
                const int trow = (e8 & 3) + 8 * (e8 >> 2);  // tile within the round, less 4 (lane >> 5)
                mwr[q * (16 * NT) + trow * NT] = acc[q][8 * round + e8];
            }
        const int tt = t0 + 16 * round + (tid2 >> 5);
        const int img = tt / tpi, rem = tt - img * tpi;
        const int sub = rem / (th * th), tq = rem - sub * th * th;
        const int a = sub / d + 3 * d * (tq / th), b = sub % d + 3 * d * (tq % th);  // pixel (0, 0) of the tile
        // (the mask values of this thread's 2 x 9 outputs are requested in front of the barrier the exchange needs anyway)
        float mk[2][9];
        if (p.mask) {  // block-uniform
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = n0 + (tid2 & 31) + 32 * h;
                const bool ok = tt < p.T && n < p.N;
                const float *mo = p.mask + ((size_t)(img * p.H + a) * p.W + b) * p.N + n;
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int i = 0; i < 3; ++i) mk[h][3 * j + i] = ok ? mo[((size_t)(d * i) * p.W + d * j) * p.N] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float z[5][3];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                float m[5];
#pragma unroll
                for (int v = 0; v < 5; ++v) m[v] = mrd[(5 * u + v) * (16 * NT) + 32 * h];
                at3(m[0], m[1], m[2], m[3], m[4], z[u][0], z[u][1], z[u][2]);
            }
            const int n = n0 + (tid2 & 31) + 32 * h;
            const bool ok = tt < p.T && n < p.N;
            float *o = p.y + ((size_t)(img * p.H + a) * p.W + b) * p.N + n;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float yv[3];
                at3(z[0][j], z[1][j], z[2][j], z[3][j], z[4][j], yv[0], yv[1], yv[2]);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    float v = yv[i] + bias2[h];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.mask) v = mk[h][3 * j + i] > 0.f ? v : 0.f;
                    if (ok) o[((size_t)(d * i) * p.W + d * j) * p.N] = v;
                }
            }
        }
    }
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino3_conv_kernel(const Wino3Params p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int xcd = blockIdx.x & 7, l_ = blockIdx.x >> 3;
    const int nb = l_ % p.nblocks;
    const int mb = (l_ / p.nblocks) * 8 + xcd;
    if (mb >= p.mblocks) return;  // block-uniform
    const int n0 = nb * f3::NT, t0 = mb * f3::MT;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // three straight-line copies selected once: the K loop never sees a branch on the wave's role
    if (wave < 4) wino3_body<6, 0, false>(p, n0, t0, lds);
    else if (wave < 6) wino3_body<6, 1, false>(p, n0, t0, lds);
    else wino3_body<7, 1, false>(p, n0, t0, lds);
}

// the same with tiles that have neighbours inside their sub-grid (Wino3Params::th > 1)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino3h_conv_kernel(const Wino3Params p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int xcd = blockIdx.x & 7, l_ = blockIdx.x >> 3;
    const int nb = l_ % p.nblocks;
    const int mb = (l_ / p.nblocks) * 8 + xcd;
    if (mb >= p.mblocks) return;  // block-uniform
    const int n0 = nb * f3::NT, t0 = mb * f3::MT;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) wino3_body<6, 0, true>(p, n0, t0, lds);
    else if (wave < 6) wino3_body<6, 1, true>(p, n0, t0, lds);
    else wino3_body<7, 1, true>(p, n0, t0, lds);
}

}  // namespace

namespace mpsr {

// winograd3w.hip: the same layer with one wave owning all 25 positions of its tile block
bool winograd3w_applies(int B, int H, int W, int C, int N, int dilation);
long long winograd3w_workgroups(int B, int N, int dilation);
int launch_winograd3w(const float *x, int B, int H, int W, int C, const float *u, const float *bias, int relu, float *y,
                      int N, int dilation, hipStream_t s, const float *mask);
// winograd3z.hip: the same layer in SIXTEEN products per tile (a one-tile sub-grid reads nothing outside itself: rank 4 per
// dimension instead of F(3,3)'s 5), one wave owning all 16 positions of its tile block
int launch_winograd3z(const float *x, int B, int H, int W, int C, const float *u, const float *bias, int relu, float *y,
                      int N, int dilation, hipStream_t s, const float *mask, float *part, size_t part_floats);
int launch_winograd3z_filter(const float *w, int N, int C, float *u, hipStream_t s);
// mpsr_debug_set_wino3_form: -1 = by size, 0 = F(3x3,3x3) with positions shared by eight waves (this file), 1 = F(3x3,3x3)
// with one wave per tile block (winograd3w.hip), 2 = the sixteen-product form (winograd3z.hip)
std::atomic<int> g_wino3_form{-1};

// which kernel serves a layer that conv3x3_winograd3 takes: 0 / 1 / 2 as above.  One tile per sub-grid: the sixteen-product
// form at EVERY batch size -- fewer products and the error of a direct convolution; below ~190 workgroups (B < 192 at
// dilation 4) its one-wave-per-SIMD workgroups do not fill the chip, so it cuts such launches along K (SPLIT
// instantiation: 32 us at B = 32 where the unsplit kernel and this file's both need ~70-76).  Tiles with halos (th > 1)
// have real neighbours: this file's kernels.
int winograd3_form(int B, int H, int W, int C, int N, int dilation)
{
    const int th = H / (3 * dilation);
    if (th != 1 || !winograd3w_applies(B, H, W, C, N, dilation)) return 0;
    const int form = g_wino3_form.load();
    return (form >= 0 && form <= 2) ? form : 2;
}
// ... and the form of the transformed filters that kernel reads (FILTER_FORM_*: the filter cache's tags, the tail job)
int winograd3_filter_form(int B, int H, int W, int C, int N, int dilation)
{
    return winograd3_form(B, H, W, C, N, dilation) == 2 ? FILTER_FORM_WINO3Z : FILTER_FORM_WINO3;
}

thread_local FilterTailJob g_filter_tail_job;
thread_local FilterTailJob g_filter_tail_done;
thread_local FilterCacheSlot g_filter_cache_slot;

size_t winograd3_scratch_floats(int C, int N) { return (size_t)f3::NP * N * C; }

// 3x3, dilation d, H = W = 3 d th (the sub-grids are 3 th x 3 th: th x th tiles of 3x3; th = 1: ResNet-101 block3 at
// output stride 4, th = 2: block2), C % 16 == 0
bool winograd3_applies(int H, int W, int C, int dilation)
{
    return dilation >= 1 && H == W && H % (3 * dilation) == 0 && H / (3 * dilation) <= 16 && C % 16 == 0 && C >= 16;
}

// tiles of a launch (one per 3x3 block of a pixel sub-grid) and the multiply-adds it issues: ONE definition for the
// launcher below and mpsr_conv2d_plan (bench.py's roofline.executed)
long long winograd3_tiles(int B, int H, int dilation)
{
    const int th = H / (3 * dilation);
    return (long long)B * dilation * dilation * th * th;
}
double winograd3_executed_flops(int B, int H, int C, int N, int dilation)
{
    const double products = winograd3_form(B, H, H, C, N, dilation) == 2 ? 16.0 : 25.0;
    return 2.0 * (double)winograd3_tiles(B, H, dilation) * products * C * N;
}

int conv3x3_winograd3(const float *x, int B, int H, int W, int C, const float *w, const float *bias, int relu,
                      float *y, int N, int dilation, float *ws, size_t ws_floats, hipStream_t s, const float *mask)
{
    using namespace f3;
    MPSR_REQUIRE(winograd3_applies(H, W, C, dilation),
                 "conv3x3_winograd3: needs H = W = a multiple of 3 * dilation and C %% 16 == 0");
    if (ws_floats < winograd3_scratch_floats(C, N) || !ws)
        return fail(MPSR_ERR_WORKSPACE, "conv3x3_winograd3: scratch holds %zu floats, needs %zu", ws_floats,
                    winograd3_scratch_floats(C, N));
    const long long xbytes = (long long)B * H * W * C * 4;
    MPSR_REQUIRE(xbytes < 0x7ff00000LL && winograd3_scratch_floats(C, N) * 4 < 0x7ff00000ULL,
                 "conv3x3_winograd3: tensor exceeds the 2 GiB this kernel's offsets address; split the batch");
    const int th = H / (3 * dilation);
    const void *kern = th > 1 ? reinterpret_cast<const void *>(wino3h_conv_kernel) : reinterpret_cast<const void *>(wino3_conv_kernel);
    MPSR_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDSF * sizeof(float))));
    const int form = winograd3_form(B, H, W, C, N, dilation);
    const int fform = form == 2 ? FILTER_FORM_WINO3Z : FILTER_FORM_WINO3;
    // the caller's filter cache (mpsr_net_opts), if the network entry point offered a slot for this layer
    float *u = ws;
    bool ready = false;
    if (g_filter_cache_slot.w == w && g_filter_cache_slot.u && g_filter_cache_slot.floats >= winograd3_scratch_floats(C, N)) {
        u = g_filter_cache_slot.u;
        ready = g_filter_cache_slot.holds(fform);
    }
    g_filter_cache_slot = FilterCacheSlot();
    // (or the filters were transformed into `u` by the tail job of the preceding pointwise launch)
    ready = ready || (g_filter_tail_done.w == w && g_filter_tail_done.u == u && g_filter_tail_done.N == N &&
                      g_filter_tail_done.C == C && g_filter_tail_done.form == (form == 2 ? 1 : 0));
    g_filter_tail_done = FilterTailJob();
    if (!ready) {
        if (form == 2) {
            if (int rc = launch_winograd3z_filter(w, N, C, u, s)) return rc;
        } else {
            const long long total = (long long)N * C;
            hipLaunchKernelGGL(wino3_filter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, N, C, u);
            MPSR_CHECK_LAUNCH("wino3_filter_kernel");
        }
    }
    // one tile per sub-grid and enough of them to give every CU a workgroup of four one-per-SIMD waves: the forms without
    // the epilogue exchange (winograd3z.hip: sixteen products; winograd3w.hip: F(3x3,3x3), identical bits to this file's)
    if (form == 2) {
        // scratch for the K-split form of small launches (winograd3z.hip): what `ws` holds behind the transformed filters
        // (all of it when they live in the caller's cache)
        float *part = ws;
        size_t part_floats = ws_floats;
        if (u == ws) {
            const size_t off = align_up(winograd3_scratch_floats(C, N), 64);
            part = ws + off;
            part_floats = ws_floats > off ? ws_floats - off : 0;
        }
        return launch_winograd3z(x, B, H, W, C, u, bias, relu, y, N, dilation, s, mask, part, part_floats);
    }
    if (form == 1) return launch_winograd3w(x, B, H, W, C, u, bias, relu, y, N, dilation, s, mask);
    Wino3Params p;
    p.x = x; p.u = u; p.bias = bias; p.y = y; p.mask = mask;
    p.B = B; p.H = H; p.W = W; p.C = C; p.N = N; p.dil = dilation;
    p.th = th;
    p.T = (int)winograd3_tiles(B, H, dilation);
    p.cblocks = C / KC;
    p.nblocks = ceil_div(N, NT);
    p.mblocks = ceil_div(p.T, MT);
    p.relu = relu;
    p.xbytes = (unsigned)xbytes;
    p.ubytes = (unsigned)(winograd3_scratch_floats(C, N) * 4);
    const long long blocks = 8LL * ceil_div(p.mblocks, 8) * p.nblocks;
    if (blocks > 0x7fffffffLL) return fail(MPSR_ERR_UNSUPPORTED, "conv3x3_winograd3: grid too large");
    if (th > 1) hipLaunchKernelGGL(wino3h_conv_kernel, dim3((unsigned)blocks), dim3(512), LDSF * sizeof(float), s, p);
    else hipLaunchKernelGGL(wino3_conv_kernel, dim3((unsigned)blocks), dim3(512), LDSF * sizeof(float), s, p);
    MPSR_CHECK_LAUNCH("wino3_conv_kernel");
    return MPSR_OK;
}

}  // namespace mpsr

extern "C" void mpsr_debug_set_wino3_form(int form) { mpsr::g_wino3_form = form; }
