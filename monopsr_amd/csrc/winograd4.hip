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
//     needs no sharing at all: wave (g, nh) is the only reader of positions g x channels nh, so its fragments go
//     straight from L2 into registers (16 bytes per lane = the k pairing of the A side; one load = 1 KiB contiguous),
//     requested one unit of 12 MFMAs ahead into the other of two register sets.  LDS carries A only.
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
#include "wino3_filter.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

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
    float *part;  // split kernel: partial outputs of the first-arriving half, laid out like y
    int *sync;    // split kernel: a ticket and a flag per pair of workgroups, zeroed before the launch
    unsigned pbytes;
    unsigned long long *trace;  // -DW4_TRACE builds: per-wave cycle stamps of two K steps (tools/wino4_trace.py)
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

#ifndef W4_BAUX
#define W4_BAUX 0  // cache policy of the transformed-filter loads / the patch loads: 2 = non-temporal (A/B builds)
#endif
#ifndef W4_AAUX
#define W4_AAUX 0
#endif
#ifndef W4_NT
#define W4_NT 0  // 1: the result leaves through non-temporal stores (A/B builds)
#endif
#ifndef W4_SPREAD
#define W4_SPREAD 0
#endif
#ifndef W4_PRIO
#define W4_PRIO 1  // static issue priority inside the K loop: 0 none, 1 waves 4-7 (shipped), 2 waves 0-3
#endif
#ifndef W4_BPRE
#define W4_BPRE 1  // B fragments requested this many units (of 12 MFMAs) ahead: 1 (two register sets) or 2 (three)
#endif
#ifndef W4_APRE
#define W4_APRE 0  // 1: read the next unit's A fragments during the current unit (12 more registers)
#endif

// IN_C8 / OUT_C8: the tensor is channel-blocked, [B][C/8][H][W][8] ("C8"), instead of NHWC.  A K step then reads
// whole 128-byte lines (4 pixels x 8 channels) instead of 32 bytes out of every pixel's line: the decoder's internal
// tensors use it (network.hip), the generic mpsr_conv2d_nhwc_f32 path does not.
template <bool IN_C8, bool OUT_C8>
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

#ifdef W4_TRACE
    unsigned long long ts[16];
    int nts = 0;
#define W4_STAMP(cond) do { if ((cond) && nts < 16) ts[nts++] = __builtin_readcyclecounter(); } while (0)
#else
#define W4_STAMP(cond) do { } while (0)
#endif
    W4_STAMP(true);  // 0: start
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
    const unsigned shift = (unsigned)(p.W + 1) * (IN_C8 ? 8u : (unsigned)p.C) * 4u;
    char *xback = const_cast<char *>(reinterpret_cast<const char *>(p.x)) - shift;
    // Only the ring of a 6x6 patch can leave the image (row 0 / 5, column 0 / 5): nine distinct per-thread byte
    // offsets (this thread's pixel (0,0) of the patch, or an out-of-range value) cover the 36 requests; the pixel
    // itself comes in through the scalar offset.  (One select per request instead would be hoisted out of the K loop
    // by hipcc as 36 registers.)
    unsigned voffc[3][3];
    {
        const int t = t0 + lt;
        const int tpi = p.th * p.tw;
        const int img = t / tpi, rem = t - img * tpi;
        const int ty = rem / p.tw, tx = rem - ty * p.tw;
        const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
        // (+ shift, i.e. relative to the moved-back descriptor this is the offset of pixel (y0, x0))
        const unsigned abase = IN_C8 ? (unsigned)(((img * (p.C / 8) * p.H + y0 + 1) * p.W + x0 + 1) * 8 + ch) * 4u
                                     : (unsigned)(((img * p.H + y0 + 1) * p.W + x0 + 1) * p.C + ch) * 4u;
        const bool in = t < p.T;
        const bool rowc[3] = {in && y0 >= 0, in, in && y0 + 5 < p.H};
        const bool colc[3] = {x0 >= 0, true, x0 + 5 < p.W};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) voffc[a][b] = (rowc[a] && colc[b]) ? abase : OOB;
    }
    float pa[36];  // the patch, then its transform, row-major
    // request number L of a patch, column by column (the column transforms start with column 0)
    auto load_patch1 = [&](int step, int L) __attribute__((always_inline)) {
        const int s = L / 6, r = L % 6;
#ifdef W4_SKIP_LOAD
        if (step > 1) return;
#endif
        const bool live = step < nsteps;
        // (a request past the last channel block goes through a zero-length descriptor: a scalar select)
        const __amdgpu_buffer_rsrc_t rr =
            __builtin_amdgcn_make_buffer_rsrc(xback, 0, live ? (int)(p.xbytes + shift) : 0, 0x00020000);
        const unsigned so = IN_C8 ? (unsigned)(((live ? step : 0) * p.H + r) * p.W + s) * 32u
                                  : (unsigned)((r * p.W + s) * p.C + (live ? step : 0) * KC) * 4u;
        pa[6 * r + s] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(rr, voffc[r == 0 ? 0 : r == 5 ? 2 : 1][s == 0 ? 0 : s == 5 ? 2 : 1],
                                                        so, W4_AAUX));
    };
    // B^T applied to six values in place, in three parts of four operations (parts 0 and 1 read the original values,
    // part 2 finishes from part 0's temporaries)
    float ta, tb, tc, te;
    auto bt_part = [&](int part, float &d0, float &d1, float &d2, float &d3, float &d4, float &d5)
                       __attribute__((always_inline)) {
#ifdef W4_SKIP_VALU
        return;
#endif
        if (part == 0) {
            ta = fmaf(-4.f, d2, d4);
            tb = fmaf(-4.f, d1, d3);
            tc = d4 - d2;
            te = d3 - d1;
        } else if (part == 1) {
            d0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
            d5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
        } else {
            d1 = ta + tb;
            d2 = ta - tb;
            d3 = fmaf(2.f, te, tc);
            d4 = fmaf(-2.f, te, tc);
        }
    };
    auto vertical = [&](int s, int part) __attribute__((always_inline)) {
        bt_part(part, pa[s], pa[6 + s], pa[12 + s], pa[18 + s], pa[24 + s], pa[30 + s]);
    };
    auto horizontal = [&](int r, int part) __attribute__((always_inline)) {
        bt_part(part, pa[6 * r], pa[6 * r + 1], pa[6 * r + 2], pa[6 * r + 3], pa[6 * r + 4], pa[6 * r + 5]);
    };
    // A[buf][pos][tile][8 channels], 16-byte halves swapped on odd 8-row blocks
    float *awr = lds + lt * 8 + 4 * ((ch >> 2) ^ ((lt >> 3) & 1)) + (ch & 3);
    auto store_a = [&](int buf, int pos) __attribute__((always_inline)) {
#ifndef W4_SKIP_STORE  // (timing experiments only: tools/README.md)
        awr[buf * ABUF + pos * APOS] = pa[pos];
#else
        asm volatile("" ::"v"(pa[pos]));
#endif
    };

    // ---- B fragments straight from the transformed filters (L2) into registers: wave (g, nh) is the only consumer
    // of its 9 positions x 32 channels, so there is nothing to share through LDS.  Lane = (n = lane & 31, k half =
    // lane >> 5): 16 bytes, one load instruction = 1 KiB contiguous.
    const unsigned bvoff =
        n0 + 32 * nh + (lane & 31) < p.N ? (unsigned)((n0 + 32 * nh + (lane & 31)) * KC + 4 * (lane >> 5)) * 4u : OOB;
    const unsigned bpstride = (unsigned)p.N * KC * 4u;  // bytes between positions of one channel block
    float4 fb[1 + W4_BPRE][3];
    auto load_b1 = [&](int step, int u, int j, int set) __attribute__((always_inline)) {
#ifdef W4_SKIP_BLOAD
        if (step > 0) return;
#endif
        const bool live = step < nsteps;
        const __amdgpu_buffer_rsrc_t rr =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, live ? (int)p.ubytes : 0, 0x00020000);
        const unsigned so = ((unsigned)(live ? step : 0) * 36u + (unsigned)(gpos + 6 * u + j)) * bpstride;
        fb[set][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, bvoff, so, W4_BAUX));
    };

    // ---- A fragment address: row = lane & 31 (tile), k half = lane >> 5
    const float *ard = lds + gpos * APOS + (lane & 31) * 8 + 4 * ((lane >> 5) ^ (((lane & 31) >> 3) & 1));
    float4 fa[1 + W4_APRE][3];
    auto read_a = [&](int buf, int u, int set) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            fa[set][j] = *reinterpret_cast<const float4 *>(ard + buf * ABUF + (6 * u + j) * APOS);
    };

    f32x16 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;

    // One K step of this wave = 3 units (u = row a of the wave's 3x3 position block) of 12 MFMAs: the unit's three
    // positions are three independent accumulators taken round-robin, k = 0..3 of the fragments outermost.  MFMA
    // number m = 12 u + 3 k + j of the step is followed by duty(m) -- the producer work of this wave -- and the whole
    // order is pinned (sched_barrier after every slot): hipcc otherwise gathers the duty arithmetic into one block
    // behind the first MFMA.  The B fragments of the NEXT unit are requested in the first three slots of a unit
    // (parity bpar of the register sets alternates unit by unit: 3 units per step, two steps per loop trip).
    auto kstep = [&](int s, auto buf_c, auto &&duty) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_c)::value;
        W4_STAMP((s >> 1) == 4);  // steps 8 and 9: step start, after each unit, after the barrier
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            // register set of this unit's B fragments: distance 1 -> the two sets alternate unit by unit (3 units per
            // step, two steps per loop trip); distance 2 -> unit u always uses set u
            const int bpar = W4_BPRE == 2 ? u : (3 * buf + u) & 1;
            const int aset = W4_APRE ? ((3 * buf + u) & 1) : 0;
            if (!W4_APRE || u == 0) read_a(buf, u, aset);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int m = 12 * u + 3 * k + j;
                    const float av = k == 0 ? fa[aset][j].x : k == 1 ? fa[aset][j].y : k == 2 ? fa[aset][j].z : fa[aset][j].w;
                    const float bv = k == 0 ? fb[bpar][j].x : k == 1 ? fb[bpar][j].y : k == 2 ? fb[bpar][j].z : fb[bpar][j].w;
                    acc[3 * u + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[3 * u + j], 0, 0, 0);
                    if (k == 0) {
                        if (W4_BPRE == 2) {  // unit u + 2 (of this step or the next) into the set unit u - 1 just left
                            if (u == 0) load_b1(s, 2, j, 2);
                            else load_b1(s + 1, u - 1, j, u - 1);
                        } else {
                            if (u < 2) load_b1(s, u + 1, j, bpar ^ 1);
                            else load_b1(s + 1, 0, j, bpar ^ 1);
                        }
                    }
                    if (W4_APRE && k == 2 && j == 0 && u < 2) read_a(buf, u + 1, aset ^ 1);
                    duty(m);
                    __builtin_amdgcn_sched_barrier(0);
                }
            W4_STAMP((s >> 1) == 4);
        }
    };
    // Producer duties.  request: 18 loads in slots 3-11 and 18 in slots 15-23, two per slot -- behind each unit's B
    // requests, so that a wait for those (the counter retires in order) never waits for younger patch loads.
    // transform: slots 0-17 the six column transforms (3 parts each), slots 18-35 the six row transforms; the stores
    // of a finished row follow two per slot from slot 21; the last row's six after the step's last MFMA.
    auto request = [&](int step) __attribute__((always_inline)) {
        return [&, step](int slot) __attribute__((always_inline)) {
#if W4_SPREAD
            load_patch1(step, slot);  // one request per slot
#else
            if (slot >= 3 && slot < 12) {
                load_patch1(step, 2 * (slot - 3));
                load_patch1(step, 2 * (slot - 3) + 1);
            } else if (slot >= 15 && slot < 24) {
                load_patch1(step, 18 + 2 * (slot - 15));
                load_patch1(step, 18 + 2 * (slot - 15) + 1);
            }
#endif
        };
    };
    auto transform = [&](auto buf_c) __attribute__((always_inline)) {
        return [&](int slot) __attribute__((always_inline)) {
            constexpr int buf = decltype(buf_c)::value;
#ifdef W4_HALF_DUTY  // timing experiment: the producer work of a workgroup that owns 18 of the 36 positions
            if (slot < 18) { if (slot % 3 != 2) vertical(slot / 3, slot % 3); }
            else if (slot < 27) horizontal((slot - 18) / 3, (slot - 18) % 3);
            if (slot >= 21 && slot < 30) {
                store_a(buf, 2 * (slot - 21));
                store_a(buf, 2 * (slot - 21) + 1);
            }
            if (true) return;
#endif
            if (slot < 18) vertical(slot / 3, slot % 3);
            else horizontal((slot - 18) / 3, (slot - 18) % 3);
            if (slot >= 21) {
                store_a(buf, 2 * (slot - 21));
                store_a(buf, 2 * (slot - 21) + 1);
            }
            if (slot == 35)
#pragma unroll
                for (int i = 30; i < 36; ++i) store_a(buf, i);
        };
    };

    // ---- prologue: B of (step 0, unit 0); A of step 0 by waves 0-3; the patch of step 1 requested by waves 4-7
#pragma unroll
    for (int j = 0; j < 3; ++j) load_b1(0, 0, j, 0);
    if (W4_BPRE == 2)
#pragma unroll
        for (int j = 0; j < 3; ++j) load_b1(0, 1, j, 1);
#pragma unroll
    for (int i = 0; i < 36; ++i) load_patch1(dgrp, i);
    if (dgrp == 0) {
#pragma unroll
        for (int k = 0; k < 18; ++k) vertical(k / 3, k % 3);
#pragma unroll
        for (int k = 0; k < 18; ++k) horizontal(k / 3, k % 3);
#pragma unroll
        for (int i = 0; i < 36; ++i) store_a(0, i);
    }
    __syncthreads();
    W4_STAMP(true);  // 1: prologue done

    // ---- K loop, two steps per trip.  Waves 0-3: even step = request the patch of step s + 2, odd step = transform
    // it into buffer 0.  Waves 4-7: even step = transform the patch of step s + 1 into buffer 1, odd step = request
    // the patch of step s + 3.
    // Static priority for the second-dispatched half of the workgroup (MI355X_MICROARCH.md, "two waves per SIMD", item 4:
    // the younger wave of a SIMD loses the issue arbitration at every segment start; one s_setprio for the whole K loop,
    // no per-segment flips).  r06, same-session step A/B over three rounds each (tools/step_libs.sh): 13.339 ms without,
    // 13.323 with waves 0-3 raised, 13.313 with waves 4-7 raised (-2.5 % of this kernel's two launches); 0 = off.
#if W4_PRIO
    if (dgrp == (W4_PRIO == 1 ? 1 : 0)) __builtin_amdgcn_s_setprio(1);
#endif
    if (dgrp == 0) {
        for (int s = 0; s < nsteps; s += 2) {
            kstep(s, IC<0>{}, request(s + 2));
            __syncthreads();
            W4_STAMP(s == 8);
            kstep(s + 1, IC<1>{}, transform(IC<0>{}));
            __syncthreads();
            W4_STAMP(s == 8);
        }
    } else {
        for (int s = 0; s < nsteps; s += 2) {
            kstep(s, IC<0>{}, transform(IC<1>{}));
            __syncthreads();
            W4_STAMP(s == 8);
            kstep(s + 1, IC<1>{}, request(s + 3));
            __syncthreads();
            W4_STAMP(s == 8);
        }
    }
    // (the 16-pass MFMA needs 18 wait states before its result is read; made explicit as in conv_mfma.hip.  Plain
    // vector registers: with an "a" constraint hipcc splits the 256-register budget 128 + 128 and spills)
#ifdef W4_TRACE
    W4_STAMP(true);  // 12: K loop done
#endif
#if W4_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    int tid2 = tid;
    asm volatile("s_nop 15\n\ts_nop 7"
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                   "+v"(acc[7]), "+v"(acc[8]), "+v"(tid2));  // (tid2: the epilogue's addresses are not hoisted above the loop)

    // ---- epilogue.  Two rounds of 16 tiles: every wave writes its 9 positions of those tiles as M[pos][tile][64 n];
    // then thread (tile = tid >> 5, n = tid & 31 and + 32) gathers the 36 positions of its two (tile, n) pairs and
    // finishes them.  Accumulator element e of a lane: tile row (e & 3) + 8 (e >> 2) + 4 (lane >> 5) of the 32,
    // output channel 32 nh + (lane & 31).
    const int lane2 = tid2 & 63;
    float *mwr = lds + gpos * (16 * 64) + (4 * (lane2 >> 5)) * 64 + 32 * nh + (lane2 & 31);
    const float *mrd = lds + (tid2 >> 5) * 64 + (tid2 & 31);
    const int tpi = p.th * p.tw;
    float bias2[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = n0 + (tid2 & 31) + 32 * h;
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
        W4_STAMP(round == 0);  // 13: round 0 exchanged
        const int tt = t0 + 16 * round + (tid2 >> 5);
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
            const int n = n0 + (tid2 & 31) + 32 * h;
            const bool ok = tt < p.T && n < p.N;
            // NHWC: 64 channels x 4 bytes contiguous per pixel; C8: 32-byte runs, the four pixels of a tile row complete
            // a 128-byte line over four consecutive stores
            const size_t pstride = OUT_C8 ? 8 : (size_t)p.N;
            float *o = OUT_C8 ? p.y + (((size_t)(img * (p.N / 8) + n / 8) * p.H + 4 * ty) * p.W + 4 * tx) * 8 + (n & 7)
                              : p.y + ((size_t)(img * p.H + 4 * ty) * p.W + 4 * tx) * p.N + n;
#pragma unroll
            for (int xx = 0; xx < 4; ++xx) {
                float yv[4];
                at4(z[0][xx], z[1][xx], z[2][xx], z[3][xx], z[4][xx], z[5][xx], yv[0], yv[1], yv[2], yv[3]);
#pragma unroll
                for (int yy = 0; yy < 4; ++yy) {
                    float v = yv[yy] + bias2[h];
                    if (p.relu) v = fmaxf(v, 0.f);
#if W4_NT
                    if (ok) __builtin_nontemporal_store(v, &o[((size_t)yy * p.W + xx) * pstride]);
#else
                    if (ok) o[((size_t)yy * p.W + xx) * pstride] = v;
#endif
                }
            }
        }
        W4_STAMP(true);  // 14, 15: round finished (stores issued)
    }
#ifdef W4_TRACE
    if (p.trace && lane2 == 0) {
        unsigned long long *dst = p.trace + ((size_t)blockIdx.x * 8 + wave) * 16;
        for (int i = 0; i < 16; ++i) dst[i] = i < nts ? ts[i] : 0;
    }
#endif
}


// ------------------------------------------------------------------------------------------------------------------
// Position-split variant (N % 128 == 0): a workgroup owns HALF of the transformed domain -- rows i = 3 half .. 3 half + 2,
// all six columns j: 18 positions -- for 32 tiles x 128 output channels.  Same 288 MFMAs per K step and wave layout
// (wave = 3 of the 6 columns x 32 of the 128 channels: 9 positions, 144 accumulators), but the producer side shrinks:
// a patch's column transforms only yield the three rows of this half (6 instead of 12 operations each), three row
// transforms instead of six, 18 stores instead of 36 -- per MFMA half the vector-ALU work and half the LDS stores of
// wino4_conv_kernel, whose 64-channel blocks recompute the whole transform per block (a knock-out of half the
// producer work measured -15 %: profiles/r03_wino4_knockouts.txt).  The output transform is linear in the rows, so
// each half produces a partial 4x4 block, Y = AT[:, rows of half 0] Z0 + AT[:, rows of half 1] Z1, and the two
// workgroups of a pair meet once, after both epilogue rounds, through a ticket: the first arriver publishes its
// partials to `part` with write-through (sc0 sc1) stores, drains them and raises a flag; the second waits for the flag
// (the first is running: it holds ticket 0), reads the partials with sc0 sc1 loads, adds its own from registers, bias,
// ReLU, and stores the result into y.  a + b == b + a, so the result does not depend on who is first.  The halves of a pair are adjacent workgroups of one XCD.  HALF is a template argument of the body (the
// kernel branches once): a run-time half put a branch into every producer slot of the K loop.
template <bool IN_C8, bool OUT_C8, int HALF>
__device__ __forceinline__ void wino4s_body(const Wino4Params &p, const int mb, const int nb, float *lds)
{
    using namespace f4;
    constexpr int ABUFS = 18 * APOS;  // one A buffer of the 18 positions (18 KB)
    constexpr int NTS = 128;
    constexpr int half = HALF;
    const int n0 = nb * NTS, t0 = mb * MT;
    const int pair = mb * p.nblocks + nb;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = wave & 3, jb = wave >> 2;  // consumer role: 32 of the 128 channels, 3 of the 6 columns
    const int dgrp = wave >> 2;               // producer role: 0 = even K steps, 1 = odd
    const int gpos = 18 * half + 3 * jb;      // global position of (row 0 of the half, first column of the wave)
    const int nsteps = p.cblocks;

    // ---- A producer: thread = (tile, channel of the step); requests exactly as in wino4_conv_kernel
    const int lt = 8 * (wave & 3) + (lane >> 3), ch = lane & 7;
    const unsigned shift = (unsigned)(p.W + 1) * (IN_C8 ? 8u : (unsigned)p.C) * 4u;
    char *xback = const_cast<char *>(reinterpret_cast<const char *>(p.x)) - shift;
    unsigned voffc[3][3];
    {
        const int t = t0 + lt;
        const int tpi = p.th * p.tw;
        const int img = t / tpi, rem = t - img * tpi;
        const int ty = rem / p.tw, tx = rem - ty * p.tw;
        const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
        const unsigned abase = IN_C8 ? (unsigned)(((img * (p.C / 8) * p.H + y0 + 1) * p.W + x0 + 1) * 8 + ch) * 4u
                                     : (unsigned)(((img * p.H + y0 + 1) * p.W + x0 + 1) * p.C + ch) * 4u;
        const bool in = t < p.T;
        const bool rowc[3] = {in && y0 >= 0, in, in && y0 + 5 < p.H};
        const bool colc[3] = {x0 >= 0, true, x0 + 5 < p.W};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) voffc[a][b] = (rowc[a] && colc[b]) ? abase : OOB;
    }
    float pa[36];  // the patch (row-major); after the transforms rows 0..2 hold the half's 18 positions
    auto load_patch1 = [&](int step, int L) __attribute__((always_inline)) {
        const int s = L / 6, r = L % 6;
        const bool live = step < nsteps;
        const __amdgpu_buffer_rsrc_t rr =
            __builtin_amdgcn_make_buffer_rsrc(xback, 0, live ? (int)(p.xbytes + shift) : 0, 0x00020000);
        const unsigned so = IN_C8 ? (unsigned)(((live ? step : 0) * p.H + r) * p.W + s) * 32u
                                  : (unsigned)((r * p.W + s) * p.C + (live ? step : 0) * KC) * 4u;
        pa[6 * r + s] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(rr, voffc[r == 0 ? 0 : r == 5 ? 2 : 1][s == 0 ? 0 : s == 5 ? 2 : 1],
                                                        so, W4_AAUX));
    };
    float ta, tb, tc, te;
    // column s of the patch -> rows 3 half .. 3 half + 2 of B^T d, left in rows 0..2; two parts of three operations
    auto vertical = [&](int s, int part) __attribute__((always_inline)) {
        float &d0 = pa[s], &d1 = pa[6 + s], &d2 = pa[12 + s], &d3 = pa[18 + s], &d4 = pa[24 + s], &d5 = pa[30 + s];
        if (half == 0) {  // t0 = 4 d0 - 5 d2 + d4;  t1, t2 = (d4 - 4 d2) +- (d3 - 4 d1)
            if (part == 0) {
                ta = fmaf(-4.f, d2, d4);
                tb = fmaf(-4.f, d1, d3);
                tc = fmaf(-5.f, d2, d4);
            } else {
                d0 = fmaf(4.f, d0, tc);
                d1 = ta + tb;
                d2 = ta - tb;
            }
        } else {  // t3, t4 = (d4 - d2) +- 2 (d3 - d1);  t5 = 4 d1 - 5 d3 + d5
            if (part == 0) {
                ta = d4 - d2;
                tb = d3 - d1;
                tc = fmaf(-5.f, d3, d5);
            } else {
                const float t5 = fmaf(4.f, d1, tc);
                d0 = fmaf(2.f, tb, ta);
                d1 = fmaf(-2.f, tb, ta);
                d2 = t5;
            }
        }
    };
    auto horizontal = [&](int r, int part) __attribute__((always_inline)) {
        float &d0 = pa[6 * r], &d1 = pa[6 * r + 1], &d2 = pa[6 * r + 2], &d3 = pa[6 * r + 3], &d4 = pa[6 * r + 4],
              &d5 = pa[6 * r + 5];
        if (part == 0) {
            ta = fmaf(-4.f, d2, d4);
            tb = fmaf(-4.f, d1, d3);
            tc = d4 - d2;
            te = d3 - d1;
        } else if (part == 1) {
            d0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
            d5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
        } else {
            d1 = ta + tb;
            d2 = ta - tb;
            d3 = fmaf(2.f, te, tc);
            d4 = fmaf(-2.f, te, tc);
        }
    };
    float *awr = lds + lt * 8 + 4 * ((ch >> 2) ^ ((lt >> 3) & 1)) + (ch & 3);
    auto store_a = [&](int buf, int lp) __attribute__((always_inline)) { awr[buf * ABUFS + lp * APOS] = pa[lp]; };

    // ---- B fragments straight from the transformed filters, lane = (n = lane & 31, k half = lane >> 5)
    const unsigned bvoff = n0 + 32 * nq + (lane & 31) < p.N
                               ? (unsigned)((n0 + 32 * nq + (lane & 31)) * KC + 4 * (lane >> 5)) * 4u : OOB;
    const unsigned bpstride = (unsigned)p.N * KC * 4u;
    float4 fb[2][3];
    auto load_b1 = [&](int step, int u, int j, int set) __attribute__((always_inline)) {
        const bool live = step < nsteps;
        const __amdgpu_buffer_rsrc_t rr =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, live ? (int)p.ubytes : 0, 0x00020000);
        const unsigned so = ((unsigned)(live ? step : 0) * 36u + (unsigned)(gpos + 6 * u + j)) * bpstride;
        fb[set][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, bvoff, so, W4_BAUX));
    };
    const float *ard = lds + 3 * jb * APOS + (lane & 31) * 8 + 4 * ((lane >> 5) ^ (((lane & 31) >> 3) & 1));
    float4 fa[3];
    auto read_a = [&](int buf, int u) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) fa[j] = *reinterpret_cast<const float4 *>(ard + buf * ABUFS + (6 * u + j) * APOS);
    };

    f32x16 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;

    // K step: as in wino4_conv_kernel (3 units = the half's 3 rows, 3 positions = the wave's 3 columns)
    auto kstep = [&](int s, auto buf_c, auto &&duty) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_c)::value;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int bpar = (3 * buf + u) & 1;
            read_a(buf, u);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int m = 12 * u + 3 * k + j;
                    const float av = k == 0 ? fa[j].x : k == 1 ? fa[j].y : k == 2 ? fa[j].z : fa[j].w;
                    const float bv = k == 0 ? fb[bpar][j].x : k == 1 ? fb[bpar][j].y : k == 2 ? fb[bpar][j].z : fb[bpar][j].w;
                    acc[3 * u + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[3 * u + j], 0, 0, 0);
                    if (k == 0) {
                        if (u < 2) load_b1(s, u + 1, j, bpar ^ 1);
                        else load_b1(s + 1, 0, j, bpar ^ 1);
                    }
                    duty(m);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    };
    // producer duties: request as before; transform: slots 0-11 the six half-column transforms (2 parts each),
    // 12-20 the three row transforms (3 parts each), the 18 stores two per slot from slot 15 (a row's six follow its
    // last part)
    auto request = [&](int step) __attribute__((always_inline)) {
        return [&, step](int slot) __attribute__((always_inline)) {
            if (slot >= 3 && slot < 12) {
                load_patch1(step, 2 * (slot - 3));
                load_patch1(step, 2 * (slot - 3) + 1);
            } else if (slot >= 15 && slot < 24) {
                load_patch1(step, 18 + 2 * (slot - 15));
                load_patch1(step, 18 + 2 * (slot - 15) + 1);
            }
        };
    };
    auto transform = [&](auto buf_c) __attribute__((always_inline)) {
        return [&](int slot) __attribute__((always_inline)) {
            constexpr int buf = decltype(buf_c)::value;
            if (slot < 12) vertical(slot / 2, slot % 2);
            else if (slot < 21) horizontal((slot - 12) / 3, (slot - 12) % 3);
            if (slot >= 15 && slot < 24) {
                store_a(buf, 2 * (slot - 15));
                store_a(buf, 2 * (slot - 15) + 1);
            }
        };
    };

    // ---- prologue
#pragma unroll
    for (int j = 0; j < 3; ++j) load_b1(0, 0, j, 0);
#pragma unroll
    for (int i = 0; i < 36; ++i) load_patch1(dgrp, i);
    if (dgrp == 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) vertical(k / 2, k % 2);
#pragma unroll
        for (int k = 0; k < 9; ++k) horizontal(k / 3, k % 3);
#pragma unroll
        for (int i = 0; i < 18; ++i) store_a(0, i);
    }
    __syncthreads();

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
    int tid2 = tid;
    asm volatile("s_nop 15\n\ts_nop 7"
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                   "+v"(acc[7]), "+v"(acc[8]), "+v"(tid2));

    // ---- epilogue.  Two rounds of 16 tiles: M[lp][tile][128 n] of the half's 18 positions through LDS (144 KB); thread
    // (tile = tid >> 5, channel quad = tid & 31) reads its 18 x 4 values (16-byte reads), forms the partial 4x4 block
    // of its four channels for both rounds (128 registers; the accumulators are dead by then), and the pair meets.
    const int lane2 = tid2 & 63;
    float *mwr = lds + 3 * jb * (16 * NTS) + (4 * (lane2 >> 5)) * NTS + 32 * nq + (lane2 & 31);
    const float *mrd = lds + (tid2 >> 5) * NTS + 4 * (tid2 & 31);
    const int tpi = p.th * p.tw;
    int *ticket_slot = reinterpret_cast<int *>(lds + LDSF);  // one word behind the exchange area
    const __amdgpu_buffer_rsrc_t rpart = __builtin_amdgcn_make_buffer_rsrc(p.part, 0, (int)p.pbytes, 0x00020000);
    const size_t pstride = OUT_C8 ? 8 : (size_t)p.N;
    const int n = n0 + 4 * (tid2 & 31);  // this thread's four channels
    f32x4 mine[2][16];                   // [round][pixel]: this half's partial outputs
    unsigned obase[2];
    bool okr[2];
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (round) __syncthreads();
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
                const int trow = (e8 & 3) + 8 * (e8 >> 2);  // tile within the round, less 4 (lane >> 5)
                mwr[(6 * (q / 3) + q % 3) * (16 * NTS) + trow * NTS] = acc[q][8 * round + e8];
            }
        __syncthreads();
        const int tt = t0 + 16 * round + (tid2 >> 5);
        const int img = tt / tpi, rem = tt - img * tpi;
        const int ty = rem / p.tw, tx = rem - ty * p.tw;
        okr[round] = tt < p.T;
        obase[round] = (unsigned)((OUT_C8 ? (((size_t)(img * (p.N / 8) + n / 8) * p.H + 4 * ty) * p.W + 4 * tx) * 8 + (n & 7)
                                          : ((size_t)(img * p.H + 4 * ty) * p.W + 4 * tx) * p.N + n) * 4);
        f32x4 z[3][4];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            f32x4 m[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) m[j] = *reinterpret_cast<const f32x4 *>(mrd + (6 * a + j) * (16 * NTS));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float z0, z1, z2, z3;
                at4(m[0][c], m[1][c], m[2][c], m[3][c], m[4][c], m[5][c], z0, z1, z2, z3);
                z[a][0][c] = z0;
                z[a][1][c] = z1;
                z[a][2][c] = z2;
                z[a][3][c] = z3;
            }
        }
#pragma unroll
        for (int xx = 0; xx < 4; ++xx) {
            // rows of A^T restricted to this half: half 0 (i = 0,1,2): [1 1 1; 0 1 -1; 0 1 1; 0 1 -1];
            // half 1 (i = 3,4,5): [1 1 0; 2 -2 0; 4 4 0; 8 -8 1]
            const f32x4 za = z[0][xx], zb = z[1][xx], zc = z[2][xx];
            f32x4 y0, y1, y2, y3;
            if (HALF == 0) {
                y0 = za + zb + zc;
                y1 = zb - zc;
                y2 = zb + zc;
                y3 = zb - zc;
            } else {
                const f32x4 sm = za + zb, df = za - zb;
                y0 = sm;
                y1 = df + df;
                y2 = sm * 4.f;
                y3 = df * 8.f + zc;
            }
            mine[round][xx] = y0;
            mine[round][4 + xx] = y1;
            mine[round][8 + xx] = y2;
            mine[round][12 + xx] = y3;
        }
    }
    // One ticket per pair.  Ticket 0: publish both rounds' partials (16-byte write-through stores), drain, raise the
    // flag, done.  Ticket 1: the partner holds ticket 0 and is publishing -- wait for its flag (one lane polls), read,
    // finish.  (Measured against "both publish, the last one finishes, nobody waits": that form writes every partial.)
    __syncthreads();
#ifdef W4S_NOSYNC  // (timing experiment: no publication, no ticket -- wrong results)
    if (tid2 == 0) *ticket_slot = 1;
#else
    if (tid2 == 0) *ticket_slot = __hip_atomic_fetch_add(p.sync + 2 * pair, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    __syncthreads();
#ifndef W4S_NOSYNC
    if (*ticket_slot == 0) {
#pragma unroll
        for (int round = 0; round < 2; ++round)
            if (okr[round])
#pragma unroll
                for (int px = 0; px < 16; ++px)  // pixel (yy, xx) = (px >> 2, px & 3)
                    __builtin_amdgcn_raw_buffer_store_b128(
                        __builtin_bit_cast(u32x4, mine[round][px]), rpart,
                        obase[round] + (unsigned)(((px >> 2) * p.W + (px & 3)) * pstride * 4), 0, 17);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every store has left before the flag does
        __syncthreads();
        if (tid2 == 0) __hip_atomic_store(p.sync + 2 * pair + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (tid2 == 0)
        while (__hip_atomic_load(p.sync + 2 * pair + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
            __builtin_amdgcn_s_sleep(2);
    __syncthreads();
#endif
    {  // the partner's partials are complete: finish the outputs
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bias4 = *reinterpret_cast<const f32x4 *>(p.bias + n);
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            if (!okr[round]) continue;
            f32x4 other[16];
#pragma unroll
            for (int px = 0; px < 16; ++px)
#ifdef W4S_NOSYNC
                other[px] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
                other[px] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                               rpart, obase[round] + (unsigned)(((px >> 2) * p.W + (px & 3)) * pstride * 4), 0, 17));
#endif
#pragma unroll
            for (int px = 0; px < 16; ++px) {
                f32x4 v = mine[round][px] + other[px] + bias4;
                if (p.relu)
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
                *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(p.y) + obase[round] +
                                           (size_t)(((px >> 2) * p.W + (px & 3)) * pstride * 4)) = v;
            }
        }
    }
}

template <bool IN_C8, bool OUT_C8>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino4s_conv_kernel(const Wino4Params p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int xcd = blockIdx.x & 7, l_ = blockIdx.x >> 3;
    const int half = l_ & 1, rest = l_ >> 1;
    const int nb = rest % p.nblocks;
    const int mb = (rest / p.nblocks) * 8 + xcd;
    if (mb >= p.mblocks) return;  // block-uniform
    if (half == 0) wino4s_body<IN_C8, OUT_C8, 0>(p, mb, nb, lds);
    else wino4s_body<IN_C8, OUT_C8, 1>(p, mb, nb, lds);
}

}  // namespace

// FETCH_SIZE calibration (tools/fetch_calibration.sh): stream a buffer exactly once, MODE 0 with 16 bytes per lane
// fully coalesced (the pattern MI355X_MICROARCH.md calibrates: the counter reports half the bytes), MODE 1 with this
// file's patch-request pattern -- 4 bytes per lane, 8 lanes = one 32-byte run, 8 runs 256 bytes apart per instruction,
// the rest of each 256-byte row by the following instructions.
namespace {
template <int MODE>
__global__ __launch_bounds__(256) void w4_fetch_calibration_kernel(const float *__restrict__ p, size_t floats,
                                                                   float *__restrict__ sink)
{
    float acc = 0.f;
    if (MODE == 0) {
        const float4 *q = reinterpret_cast<const float4 *>(p);
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < floats / 4; i += (size_t)gridDim.x * 256) {
            const float4 v = q[i];
            acc += v.x + v.y + v.z + v.w;
        }
    } else {
        const int lane = threadIdx.x & 63, tl = lane >> 3, ch = lane & 7;
        const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
        for (size_t blk = wave; blk * 512 + 512 <= floats; blk += nwaves)  // 8 rows x 64 floats per block
#pragma unroll
            for (int chunk = 0; chunk < 8; ++chunk) acc += p[blk * 512 + (size_t)tl * 64 + chunk * 8 + ch];
    }
    if (acc == 12345.678f) sink[0] = acc;
}
}  // namespace
extern "C" int mpsr_debug_fetch_calibration(const float *p, size_t floats, int mode, float *sink, mpsr_stream_t stream)
{
    MPSR_REQUIRE(p && sink && floats >= 512 && (mode == 0 || mode == 1), "fetch_calibration: bad arguments");
    if (mode == 0)
        hipLaunchKernelGGL(w4_fetch_calibration_kernel<0>, dim3(4096), dim3(256), 0, mpsr::as_stream(stream), p, floats, sink);
    else
        hipLaunchKernelGGL(w4_fetch_calibration_kernel<1>, dim3(4096), dim3(256), 0, mpsr::as_stream(stream), p, floats, sink);
    MPSR_CHECK_LAUNCH("w4_fetch_calibration_kernel");
    return MPSR_OK;
}

static unsigned long long *g_wino4_trace = nullptr;
extern "C" void mpsr_debug_set_wino4_trace(void *buf) { g_wino4_trace = static_cast<unsigned long long *>(buf); }

namespace mpsr { extern std::atomic<int> g_wino4_split; }
extern "C" void mpsr_debug_set_wino4_split(int on) { mpsr::g_wino4_split = on; }

namespace mpsr {

// floats of scratch conv3x3_winograd4 needs behind `ws` (the transformed filters)
size_t winograd4_scratch_floats(int C, int N) { return (size_t)36 * N * C; }

bool winograd4_applies(int H, int W, int C, int N)
{
    return H % 4 == 0 && W % 4 == 0 && C % 16 == 0 && C >= 16 && N >= 1;
}

// 0 (default): never; 1: the position-split kernel where N % 128 == 0 and its scratch fits.  Off by default: its K loop
// is 9 % faster, the hand-over of the partial outputs between the two workgroups of a pair (131 KB, write-through,
// ~6 us publish + ~2 us read: MI355X_MICROARCH.md publish-large / handoff-payload) costs 6 % -- 3.06 vs 3.16 ms for the
// decoder's four layers, not worth a spin-wait in the default path (DESIGN.md 4.1 d)
std::atomic<int> g_wino4_split{0};

// floats behind `part` (or behind the filters in `ws`) the position-split kernel needs: the first arrivers' partial
// outputs (laid out like y) + per (pair of workgroups, round) a ticket and a flag
size_t winograd4_split_floats(int B, int H, int W, int N)
{
    const size_t pairs = (size_t)ceil_div(B * (H / 4) * (W / 4), f4::MT) * (size_t)ceil_div(N, 128);
    return align_up((size_t)B * H * W * N, 64) + 2 * pairs;
}

int conv3x3_winograd4(const float *x, int B, int H, int W, int C, const float *w, const float *bias, int relu,
                      float *y, int N, float *ws, size_t ws_floats, hipStream_t s, int in_c8, int out_c8, float *part,
                      size_t part_floats)
{
    using namespace f4;
    MPSR_REQUIRE(winograd4_applies(H, W, C, N), "conv3x3_winograd4: needs H, W multiples of 4 and C %% 16 == 0");
    MPSR_REQUIRE(!out_c8 || N % 8 == 0, "conv3x3_winograd4: a channel-blocked output needs N %% 8 == 0 (N=%d)", N);
    if (ws_floats < winograd4_scratch_floats(C, N) || !ws)
        return fail(MPSR_ERR_WORKSPACE, "conv3x3_winograd4: scratch holds %zu floats, needs %zu", ws_floats,
                    winograd4_scratch_floats(C, N));
    const long long xbytes = (long long)B * H * W * C * 4;
    MPSR_REQUIRE(xbytes + (long long)(W + 1) * C * 4 < 0x7ff00000LL && winograd4_scratch_floats(C, N) * 4 < 0x7ff00000ULL,
                 "conv3x3_winograd4: tensor exceeds the 2 GiB this kernel's offsets address; split the batch");
    // position-split kernel: N a multiple of 128 and room for the partial outputs -- the caller's `part`, or the rest
    // of `ws` behind the transformed filters
    const size_t ufl = align_up(winograd4_scratch_floats(C, N), 64), need = winograd4_split_floats(B, H, W, N);
    const long long obytes = (long long)B * H * W * N * 4;
    bool split = g_wino4_split.load() != 0 && N % 128 == 0 && obytes < 0x7ff00000LL;
    if (split && !(part && part_floats >= need)) {
        if (ws_floats >= ufl + need) {
            part = ws + ufl;
            part_floats = ws_floats - ufl;
        } else {
            split = false;
        }
    }
    // the caller's filter cache (mpsr_net_opts), if the network entry point offered a slot for this layer
    float *u = ws;
    bool ready = false;
    if (g_filter_cache_slot.w == w && g_filter_cache_slot.u && g_filter_cache_slot.floats >= winograd4_scratch_floats(C, N)) {
        u = g_filter_cache_slot.u;
        ready = g_filter_cache_slot.holds(FILTER_FORM_WINO4);
    }
    g_filter_cache_slot = FilterCacheSlot();
    if (!ready) {
        const long long total = (long long)N * C;
        hipLaunchKernelGGL(wino4_filter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, N, C, u);
        MPSR_CHECK_LAUNCH("wino4_filter_kernel");
    }
    Wino4Params p;
    p.x = x; p.u = u; p.bias = bias; p.y = y;
    p.B = B; p.H = H; p.W = W; p.C = C; p.N = N;
    p.th = H / 4; p.tw = W / 4;
    p.T = B * p.th * p.tw;
    p.cblocks = C / KC;
    p.nblocks = ceil_div(N, split ? 128 : NT);
    p.mblocks = ceil_div(p.T, MT);
    p.relu = relu;
    p.xbytes = (unsigned)xbytes;
    p.ubytes = (unsigned)(winograd4_scratch_floats(C, N) * 4);
    p.part = nullptr; p.sync = nullptr; p.pbytes = 0;
    p.trace = g_wino4_trace;
    long long blocks = 8LL * ceil_div(p.mblocks, 8) * p.nblocks;
    if (split) blocks *= 2;
    if (blocks > 0x7fffffffLL) return fail(MPSR_ERR_UNSUPPORTED, "conv3x3_winograd4: grid too large");
    const dim3 grid((unsigned)blocks), block(512);
    if (split) {
        const size_t pairs = (size_t)p.mblocks * p.nblocks;
        p.part = part;
        p.sync = reinterpret_cast<int *>(part + align_up((size_t)B * H * W * N, 64));
        p.pbytes = (unsigned)obytes;
        MPSR_CHECK_HIP(hipMemsetAsync(p.sync, 0, 2 * pairs * sizeof(int), s));
        const size_t lds = LDSF * sizeof(float) + 16;
        const void *kfn = in_c8 ? (out_c8 ? reinterpret_cast<const void *>(wino4s_conv_kernel<true, true>)
                                          : reinterpret_cast<const void *>(wino4s_conv_kernel<true, false>))
                                : (out_c8 ? reinterpret_cast<const void *>(wino4s_conv_kernel<false, true>)
                                          : reinterpret_cast<const void *>(wino4s_conv_kernel<false, false>));
        MPSR_CHECK_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (in_c8 && out_c8) hipLaunchKernelGGL((wino4s_conv_kernel<true, true>), grid, block, lds, s, p);
        else if (in_c8) hipLaunchKernelGGL((wino4s_conv_kernel<true, false>), grid, block, lds, s, p);
        else if (out_c8) hipLaunchKernelGGL((wino4s_conv_kernel<false, true>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((wino4s_conv_kernel<false, false>), grid, block, lds, s, p);
        MPSR_CHECK_LAUNCH("wino4s_conv_kernel");
        return MPSR_OK;
    }
    // (set on every call: cheap, idempotent, and right for whichever device is current)
    const void *kfn = in_c8 ? (out_c8 ? reinterpret_cast<const void *>(wino4_conv_kernel<true, true>)
                                      : reinterpret_cast<const void *>(wino4_conv_kernel<true, false>))
                            : (out_c8 ? reinterpret_cast<const void *>(wino4_conv_kernel<false, true>)
                                      : reinterpret_cast<const void *>(wino4_conv_kernel<false, false>));
    MPSR_CHECK_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDSF * sizeof(float))));
    if (in_c8 && out_c8) hipLaunchKernelGGL((wino4_conv_kernel<true, true>), grid, block, LDSF * sizeof(float), s, p);
    else if (in_c8) hipLaunchKernelGGL((wino4_conv_kernel<true, false>), grid, block, LDSF * sizeof(float), s, p);
    else if (out_c8) hipLaunchKernelGGL((wino4_conv_kernel<false, true>), grid, block, LDSF * sizeof(float), s, p);
    else hipLaunchKernelGGL((wino4_conv_kernel<false, false>), grid, block, LDSF * sizeof(float), s, p);
    MPSR_CHECK_LAUNCH("wino4_conv_kernel");
    return MPSR_OK;
}

}  // namespace mpsr
