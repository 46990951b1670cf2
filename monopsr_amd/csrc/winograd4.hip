// 3x3 stride-1 SAME convolution by Winograd F(4x4, 3x3) on fp32 MFMA: the map decoder's four dense 3x3 layers
// (reference graph: monopsr/builders/net_builder.py:70-89).
//
//   Y (4x4 block of output pixels) = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A,     d = the 6x6 input patch around it.
// 36 products per (channel pair, 16 output pixels) where the direct form has 144 and F(2x2,3x3) (winograd.hip) 64:
// the matrix pipes issue 1/4 of the direct multiply-adds.  The 36 element positions are 36 independent GEMMs
//   M_p[tile][n] = sum_c V_p[tile][c] * U_p[n][c]           (tile = one 4x4 output block, p = 6 i + j).
//
// MI355X design -- one kernel does input transform, the 36 GEMMs and the output transform; V and M never reach HBM.
//   * workgroup = 32 tiles x 64 output channels x all 36 positions, 512 threads = two waves per SIMD.  Wave (g, nh):
//     g = (ib, jb) owns the 3x3 block of positions i in 3 ib .. 3 ib + 2, j in 3 jb .. 3 jb + 2; nh = which 32 of the 64
//     channels.  9 accumulator tiles of v_mfma_f32_32x32x2_f32 = 144 accumulator registers per wave.
//   * K step = 8 input channels.  LDS rows are 8 floats = two 16-byte halves, XOR-swizzled by bit 3 of the row so that
//     ds_read_b128 fragments (row = lane & 31, half = lane >> 5) are bank-conflict free without padding; one read
//     feeds four MFMAs (k = j and 4 + j of the step in MFMA j, the same pairing on both operands).
//   * A side (transformed patches) is DOUBLE buffered (2 x 36 KB).  A thread owns one (tile, channel) of a step: 36
//     four-byte loads of its 6x6 patch (8 lanes = 32 contiguous bytes per pixel; out-of-image pixels by an
//     out-of-range offset), B^T d B in place (144 operations), 36 ds_write_b32.  32 tiles x 8 channels = 256 threads
//     per step, so waves 0-3 produce the even steps and waves 4-7 the odd ones; a wave alternates a "request" step (its
//     36 loads woven between its MFMAs) and a "transform" step (arithmetic + stores woven in), offset by one step
//     between the two halves -- the two loops are separate straight-line copies selected once, so the accumulators
//     never meet a branch inside the K loop.
//   * B side (transformed filters U, made per call by wino4_filter_kernel, laid out [channel block][position][n][8])
//     needs no sharing at all: wave (g, nh) is the only reader of positions g x channels nh, so it copies exactly its
//     own 9 x 32 rows through a PRIVATE single-buffered LDS region (a rolling window of three 16-byte loads in
//     flight, written back right after the MFMAs that consumed the old contents): no barrier ever guards B.
//   * one s_barrier per K step (the A hand-over).
//   * epilogue: the 36 position accumulators of a (tile, channel) live in four waves; they are exchanged through
//     LDS (all 144 KB, two rounds of 16 tiles), each thread then runs A^T M A for two (tile, channel) pairs (100
//     operations each), adds bias / ReLU and stores 64 channels x 4 bytes contiguous per pixel.
// Numerics: fp32 throughout; transform constants up to 8 and 1/24: measured error against float64 ~1.5e-5 of the
// tensor scale (F(2x2): 1e-6, direct: 5e-7) -- inside the path's 1e-3 budget, tests/test_net_gpu.py keeps <= 1e-4.
#include <atomic>
#include <mutex>
#include <type_traits>

#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace f4 {
constexpr int MT = 32, NT = 64, KC = 8;
constexpr int APOS = MT * KC;          // floats per position of an A buffer (256)
constexpr int BPOS = NT * KC;          // floats per position of the B region (512)
constexpr int ABUF = 36 * APOS;        // one A buffer (9216 floats = 36 KB)
constexpr int BOFF = 2 * ABUF;         // B region starts after the two A buffers
constexpr int LDSF = BOFF + 36 * BPOS; // 36864 floats = 144 KB
constexpr unsigned OOB = 0x80000000u;  // a byte offset no tensor of this kernel reaches
}  // namespace f4

struct Wino4Params {
    const float *x, *u, *bias;
    float *y;
    int B, H, W, C, N, th, tw, T;  // th x tw tiles per image, T tiles in all
    int cblocks, nblocks, mblocks, relu;
    unsigned xbytes, ubytes;
};

// U[cb][pos][n][8] = (G g G^T)[pos] for filter g = w[n][(ky*3+kx)*C + c], c = cb*8 + j.  One thread per (n, c).
__global__ __launch_bounds__(256) void wino4_filter_kernel(const float *__restrict__ w, int N, int C, float *__restrict__ u)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * C) return;
    const int n = (int)(i / C), c = (int)(i - (long long)n * C);
    double g[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = (double)w[(size_t)n * 9 * C + (size_t)(ky * 3 + kx) * C + c];
    auto gcol = [](double a, double b, double c2, double *o) {  // G (6x3) applied to one 3-vector
        o[0] = a / 4.0;
        o[1] = -(a + b + c2) / 6.0;
        o[2] = -(a - b + c2) / 6.0;
        o[3] = a / 24.0 + b / 12.0 + c2 / 6.0;
        o[4] = a / 24.0 - b / 12.0 + c2 / 6.0;
        o[5] = c2;
    };
    double t[6][3];  // G g
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        double o[6];
        gcol(g[0][kx], g[1][kx], g[2][kx], o);
#pragma unroll
        for (int r = 0; r < 6; ++r) t[r][kx] = o[r];
    }
    float *dst = u + ((size_t)(c / f4::KC) * 36 * N + n) * f4::KC + (c % f4::KC);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        double o[6];
        gcol(t[r][0], t[r][1], t[r][2], o);
#pragma unroll
        for (int s = 0; s < 6; ++s) dst[(size_t)(r * 6 + s) * N * f4::KC] = (float)o[s];
    }
}

// B^T (6x6) applied to a 6-vector, in place: 12 operations
__device__ __forceinline__ void bt6(float &d0, float &d1, float &d2, float &d3, float &d4, float &d5)
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
// A^T (4x6) applied to a 6-vector: 10 operations
__device__ __forceinline__ void at4(float m0, float m1, float m2, float m3, float m4, float m5, float &z0, float &z1,
                                    float &z2, float &z3)
{
    const float p = m1 + m2, q = m1 - m2, r = m3 + m4, t = m3 - m4;
    z0 = m0 + p + r;
    z1 = fmaf(2.f, t, q);
    z2 = fmaf(4.f, r, p);
    z3 = fmaf(8.f, t, q) + m5;
}

template <int V>
using IC = std::integral_constant<int, V>;

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino4_conv_kernel(const Wino4Params p)
{
    using namespace f4;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    // XCD x (workgroup b runs on XCD b % 8: speed only) takes M blocks x, x+8, ...; the N blocks of one M block run
    // back to back on it, so the patches are fetched from HBM once
    const int xcd = blockIdx.x & 7, l_ = blockIdx.x >> 3;
    const int nb = l_ % p.nblocks;
    const int mb = (l_ / p.nblocks) * 8 + xcd;
    if (mb >= p.mblocks) return;  // block-uniform
    const int n0 = nb * NT, t0 = mb * MT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 1, nh = wave & 1;  // consumer role: position block, channel half
    const int ib = g >> 1, jb = g & 1;
    const int gpos = 18 * ib + 3 * jb;       // first position of the block; the block's positions are gpos + 6 a + b
    const int dgrp = wave >> 2;              // producer role: 0 = even K steps, 1 = odd
    const int nsteps = p.cblocks;

    // ---- A producer: thread = (tile, channel of the step)
    const int lt = 8 * (wave & 3) + (lane >> 3), ch = lane & 7;
    // descriptor moved back by one row + one pixel so that the per-thread base offset is never negative
    const unsigned shift = (unsigned)(p.W + 1) * (unsigned)p.C * 4u;
    char *xback = const_cast<char *>(reinterpret_cast<const char *>(p.x)) - shift;
    unsigned abase;
    bool rowok[6], colok[6];
    {
        const int t = t0 + lt;
        const int tpi = p.th * p.tw;
        const int img = t / tpi, rem = t - img * tpi;
        const int ty = rem / p.tw, tx = rem - ty * p.tw;
        const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
        abase = (unsigned)(((img * p.H + y0 + 1) * p.W + x0 + 1) * p.C + ch) * 4u;  // + shift, i.e. of pixel (y0+1, x0+1)
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            rowok[r] = t < p.T && y0 + r >= 0 && y0 + r < p.H;
            colok[r] = x0 + r >= 0 && x0 + r < p.W;
        }
    }
    float pa[36];  // the patch, then its transform, row-major
    auto load_patch1 = [&](int step, int i) __attribute__((always_inline)) {
        const int r = i / 6, s = i % 6;
        const bool live = step < nsteps;
        // (a request past the last channel block goes through a zero-length descriptor: a scalar select)
        const __amdgpu_buffer_rsrc_t rr =
            __builtin_amdgcn_make_buffer_rsrc(xback, 0, live ? (int)(p.xbytes + shift) : 0, 0x00020000);
        const unsigned so = (unsigned)((r * p.W + s) * p.C + (live ? step : 0) * KC) * 4u;
        pa[i] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(rr, (rowok[r] && colok[s]) ? abase : OOB, so, 0));
    };
    auto vertical = [&](int s) __attribute__((always_inline)) { bt6(pa[s], pa[6 + s], pa[12 + s], pa[18 + s], pa[24 + s], pa[30 + s]); };
    auto horizontal = [&](int r) __attribute__((always_inline)) {
        bt6(pa[6 * r], pa[6 * r + 1], pa[6 * r + 2], pa[6 * r + 3], pa[6 * r + 4], pa[6 * r + 5]);
    };
    // A[buf][pos][tile][8 channels], 16-byte halves swapped on odd 8-row blocks
    float *awr = lds + lt * 8 + 4 * ((ch >> 2) ^ ((lt >> 3) & 1)) + (ch & 3);
    auto store_a = [&](int buf, int pos) __attribute__((always_inline)) { awr[buf * ABUF + pos * APOS] = pa[pos]; };

    // ---- B copy (private to the wave): row 32 nh + (lane >> 1), half lane & 1 of each of its 9 positions
    const __amdgpu_buffer_rsrc_t ru =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, (int)p.ubytes, 0x00020000);
    const int brow = 32 * nh + (lane >> 1);
    const unsigned bvoff = n0 + brow < p.N ? (unsigned)((n0 + brow) * KC + 4 * (lane & 1)) * 4u : OOB;
    const unsigned bpstride = (unsigned)p.N * KC * 4u;  // bytes between positions of one channel block
    float *bwr = lds + BOFF + gpos * BPOS + brow * 8 + 4 * ((lane & 1) ^ ((brow >> 3) & 1));
    float4 bst[3];
    auto load_b1 = [&](int step, int q, int slot) __attribute__((always_inline)) {  // position q = 3 a + b of the wave's block, for K step `step`
        const bool live = step < nsteps;
        const __amdgpu_buffer_rsrc_t rr =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, live ? (int)p.ubytes : 0, 0x00020000);
        const unsigned so = ((unsigned)(live ? step : 0) * 36u + (unsigned)(gpos + 6 * (q / 3) + q % 3)) * bpstride;
        bst[slot] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, bvoff, so, 0));
    };
    auto store_b = [&](int q, int slot) __attribute__((always_inline)) {
        *reinterpret_cast<float4 *>(bwr + (6 * (q / 3) + q % 3) * BPOS) = bst[slot];
    };

    // ---- consumer fragment addresses
    const float *ard = lds + gpos * APOS + (lane & 31) * 8 + 4 * ((lane >> 5) ^ (((lane & 31) >> 3) & 1));
    const int frow = 32 * nh + (lane & 31);
    const float *brd = lds + BOFF + gpos * BPOS + frow * 8 + 4 * ((lane >> 5) ^ ((frow >> 3) & 1));

    f32x16 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;

    // One K step of this wave: 9 positions x 4 MFMAs, positions in interleaved pairs (two independent accumulators
    // alternate); after MFMA number m (0..35) comes duty(m); after a position's MFMAs its B rows are replaced by the
    // next step's and the load three positions further on is requested.
    auto kstep = [&](int s, auto buf_c, auto &&duty) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_c)::value;
        auto frag_a = [&](int q) __attribute__((always_inline)) {
            return *reinterpret_cast<const float4 *>(ard + buf * ABUF + (6 * (q / 3) + q % 3) * APOS);
        };
        auto frag_b = [&](int q) __attribute__((always_inline)) { return *reinterpret_cast<const float4 *>(brd + (6 * (q / 3) + q % 3) * BPOS); };
        auto refill = [&](int q) __attribute__((always_inline)) {
            store_b(q, q % 3);
            if (q + 3 < 9) load_b1(s + 1, q + 3, q % 3);
            else load_b1(s + 2, q + 3 - 9, q % 3);
        };
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int q0 = 2 * pp, q1 = q0 + 1;
            __builtin_amdgcn_sched_barrier(0);  // (hipcc otherwise hoists every fragment read of the step to its top)
            const float4 a0 = frag_a(q0), b0 = frag_b(q0), a1 = frag_a(q1), b1 = frag_b(q1);
            acc[q0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[q0], 0, 0, 0);
            duty(8 * pp + 0);
            acc[q1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc[q1], 0, 0, 0);
            duty(8 * pp + 1);
            acc[q0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[q0], 0, 0, 0);
            duty(8 * pp + 2);
            acc[q1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc[q1], 0, 0, 0);
            duty(8 * pp + 3);
            acc[q0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[q0], 0, 0, 0);
            duty(8 * pp + 4);
            acc[q1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc[q1], 0, 0, 0);
            duty(8 * pp + 5);
            acc[q0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[q0], 0, 0, 0);
            duty(8 * pp + 6);
            acc[q1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc[q1], 0, 0, 0);
            duty(8 * pp + 7);
            refill(q0);
            refill(q1);
        }
        {
            __builtin_amdgcn_sched_barrier(0);
            const float4 a0 = frag_a(8), b0 = frag_b(8);
            acc[8] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[8], 0, 0, 0);
            duty(32);
            acc[8] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[8], 0, 0, 0);
            duty(33);
            acc[8] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[8], 0, 0, 0);
            duty(34);
            acc[8] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[8], 0, 0, 0);
            duty(35);
            refill(8);
        }
    };
    // producer duties of a step.  request: one patch load per slot.  transform: slots 0-5 the six column transforms,
    // then per row r (slots 6 + 5 r ..): the row transform and its six stores.
    auto request = [&](int step) __attribute__((always_inline)) {
        return [&, step](int slot) __attribute__((always_inline)) { load_patch1(step, slot); };
    };
    auto transform = [&](auto buf_c) __attribute__((always_inline)) {
        return [&](int slot) __attribute__((always_inline)) {
            constexpr int buf = decltype(buf_c)::value;
            if (slot < 6) {
                vertical(slot);
            } else {
                const int r = (slot - 6) / 5, k = (slot - 6) % 5;
                if (k == 0) horizontal(r);
                else if (k == 1) { store_a(buf, 6 * r + 0); store_a(buf, 6 * r + 1); }
                else if (k == 2) { store_a(buf, 6 * r + 2); store_a(buf, 6 * r + 3); }
                else if (k == 3) store_a(buf, 6 * r + 4);
                else store_a(buf, 6 * r + 5);
            }
        };
    };

    // ---- prologue: B of step 0 for this wave; A of step 0 by waves 0-3; the patch of step 1 requested by waves 4-7
    {
        float4 b0[9];
#pragma unroll
        for (int q = 0; q < 9; ++q)
            b0[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                   ru, bvoff, (unsigned)(gpos + 6 * (q / 3) + q % 3) * bpstride, 0));
#pragma unroll
        for (int i = 0; i < 36; ++i) load_patch1(dgrp, i);
#pragma unroll
        for (int q = 0; q < 9; ++q) *reinterpret_cast<float4 *>(bwr + (6 * (q / 3) + q % 3) * BPOS) = b0[q];
    }
    if (dgrp == 0) {
#pragma unroll
        for (int s = 0; s < 6; ++s) vertical(s);
#pragma unroll
        for (int r = 0; r < 6; ++r) horizontal(r);
#pragma unroll
        for (int i = 0; i < 36; ++i) store_a(0, i);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) load_b1(1, q, q);
    __syncthreads();

    // ---- K loop, two steps per trip.  Waves 0-3: even step = request the patch of step s + 2, odd step = transform
    // it into buffer 0.  Waves 4-7: even step = transform the patch of step s + 1 into buffer 1, odd step = request
    // the patch of step s + 3.
    if (dgrp == 0) {
        for (int s = 0; s < nsteps; s += 2) {
            kstep(s, IC<0>{}, request(s + 2));
            __syncthreads();
            kstep(s + 1, IC<1>{}, transform(IC<0>{}));
            __syncthreads();
        }
    } else {
        for (int s = 0; s < nsteps; s += 2) {
            kstep(s, IC<0>{}, transform(IC<1>{}));
            __syncthreads();
            kstep(s + 1, IC<1>{}, request(s + 3));
            __syncthreads();
        }
    }
    // (the 16-pass MFMA needs 18 wait states before its result is read; made explicit as in conv_mfma.hip)
    asm volatile("s_nop 15\n\ts_nop 7"
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                   "+v"(acc[7]), "+v"(acc[8]));

    // ---- epilogue.  Two rounds of 16 tiles: every wave writes its 9 positions of those tiles as M[pos][tile][64 n];
    // then thread (tile = tid >> 5, n = tid & 31 and + 32) gathers the 36 positions of its two (tile, n) pairs and
    // finishes them.  Accumulator element e of a lane: tile row (e & 3) + 8 (e >> 2) + 4 (lane >> 5) of the 32,
    // output channel 32 nh + (lane & 31).
    float *mwr = lds + gpos * (16 * 64) + (4 * (lane >> 5)) * 64 + 32 * nh + (lane & 31);
    const float *mrd = lds + (tid >> 5) * 64 + (tid & 31);
    const int tpi = p.th * p.tw;
    float bias2[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = n0 + (tid & 31) + 32 * h;
        bias2[h] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        // (the K loop's last barrier, or the previous round's reads, precede these writes)
        if (round) __syncthreads();
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
                const int e = 8 * round + e8;
                const int trow = (e & 3) + 8 * ((e >> 2) & 1);  // tile within the round, less 4 (lane >> 5)
                mwr[(6 * (q / 3) + q % 3) * (16 * 64) + trow * 64] = acc[q][e];
            }
        __syncthreads();
        const int tt = t0 + 16 * round + (tid >> 5);
        const int img = tt / tpi, rem = tt - img * tpi;
        const int ty = rem / p.tw, tx = rem - ty * p.tw;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float z[6][4];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                float m[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) m[j] = mrd[(6 * i + j) * (16 * 64) + 32 * h];
                at4(m[0], m[1], m[2], m[3], m[4], m[5], z[i][0], z[i][1], z[i][2], z[i][3]);
            }
            const int n = n0 + (tid & 31) + 32 * h;
            const bool ok = tt < p.T && n < p.N;
            float *o = p.y + ((size_t)(img * p.H + 4 * ty) * p.W + 4 * tx) * p.N + n;
#pragma unroll
            for (int xx = 0; xx < 4; ++xx) {
                float yv[4];
                at4(z[0][xx], z[1][xx], z[2][xx], z[3][xx], z[4][xx], z[5][xx], yv[0], yv[1], yv[2], yv[3]);
#pragma unroll
                for (int yy = 0; yy < 4; ++yy) {
                    float v = yv[yy] + bias2[h];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (ok) o[((size_t)yy * p.W + xx) * p.N] = v;
                }
            }
        }
    }
}

}  // namespace

namespace mpsr {

// floats of scratch conv3x3_winograd4 needs behind `ws` (the transformed filters)
size_t winograd4_scratch_floats(int C, int N) { return (size_t)36 * N * C; }

bool winograd4_applies(int H, int W, int C, int N)
{
    return H % 4 == 0 && W % 4 == 0 && C % 16 == 0 && C >= 16 && N >= 1;
}

int conv3x3_winograd4(const float *x, int B, int H, int W, int C, const float *w, const float *bias, int relu,
                      float *y, int N, float *ws, size_t ws_floats, hipStream_t s)
{
    using namespace f4;
    MPSR_REQUIRE(winograd4_applies(H, W, C, N), "conv3x3_winograd4: needs H, W multiples of 4 and C %% 16 == 0");
    if (ws_floats < winograd4_scratch_floats(C, N) || !ws)
        return fail(MPSR_ERR_WORKSPACE, "conv3x3_winograd4: scratch holds %zu floats, needs %zu", ws_floats,
                    winograd4_scratch_floats(C, N));
    const long long xbytes = (long long)B * H * W * C * 4;
    MPSR_REQUIRE(xbytes + (long long)(W + 1) * C * 4 < 0x7ff00000LL && winograd4_scratch_floats(C, N) * 4 < 0x7ff00000ULL,
                 "conv3x3_winograd4: tensor exceeds the 2 GiB this kernel's offsets address; split the batch");
    // (set on every call: cheap, idempotent, and right for whichever device is current)
    MPSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_conv_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDSF * sizeof(float))));
    {
        const long long total = (long long)N * C;
        hipLaunchKernelGGL(wino4_filter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, N, C, ws);
        MPSR_CHECK_LAUNCH("wino4_filter_kernel");
    }
    Wino4Params p;
    p.x = x; p.u = ws; p.bias = bias; p.y = y;
    p.B = B; p.H = H; p.W = W; p.C = C; p.N = N;
    p.th = H / 4; p.tw = W / 4;
    p.T = B * p.th * p.tw;
    p.cblocks = C / KC;
    p.nblocks = ceil_div(N, NT);
    p.mblocks = ceil_div(p.T, MT);
    p.relu = relu;
    p.xbytes = (unsigned)xbytes;
    p.ubytes = (unsigned)(winograd4_scratch_floats(C, N) * 4);
    const long long blocks = 8LL * ceil_div(p.mblocks, 8) * p.nblocks;
    if (blocks > 0x7fffffffLL) return fail(MPSR_ERR_UNSUPPORTED, "conv3x3_winograd4: grid too large");
    hipLaunchKernelGGL(wino4_conv_kernel, dim3((unsigned)blocks), dim3(512), LDSF * sizeof(float), s, p);
    MPSR_CHECK_LAUNCH("wino4_conv_kernel");
    return MPSR_OK;
}

}  // namespace mpsr
