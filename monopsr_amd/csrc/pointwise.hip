// 1x1 convolution (pointwise GEMM) of the trunk's wide layers on fp32 MFMA:
//     y[m][n] = act( sum_k x[m][k] w[n][k] + bias[n] + residual[m][n] ),   m = pixel (NHWC row), K = C.
// ResNet-101 block3's conv1 / conv3, the projection shortcuts and the squash layers (reference graph
// object_detection/nets/resnet_v1.py:104-131 bottleneck(), monopsr/core/feature_extractors/resnet.py) -- 9 of the
// step's 16 ms went through the general implicit GEMM (conv_mfma.hip), whose tiles pass BOTH operands through LDS
// (one ds_write_b128 per 16-byte piece, 20-45 cycles of matrix-pipe time each: profiles/r03_mfma_with_lds_stores.txt)
// and re-read them as 2-4 fragments per 4-12 MFMAs.  This kernel is built from what the Winograd kernels measured:
//
//   * workgroup = 4 waves = 288 rows x 128 columns; wave = 9 row tiles (32 rows each) x 32 columns: 144 accumulator
//     registers, 10 fragments (9 A + 1 B) per 36 MFMAs.  M = batch x 144 pixels on the 12x12 trunk maps: 288-row
//     groups divide it exactly, and at batch 256 the 128 groups x N / 128 column blocks are 1, 2 or 4 whole rounds of
//     the 256 CUs x 2 resident workgroups -- no tail.
//   * A (activations) is shared by the four waves: 32 k per stage, 36 KB, two stages in LDS (128-byte rows, the
//     16-byte pieces XOR-swizzled by (row >> 1) & 7: ds_read_b128 fragment reads and ds_write_b128 fills are
//     conflict-free).  Every thread moves nine 16-byte pieces per stage: global -> registers a whole stage ahead,
//     registers -> LDS early in the next stage (one store per 16 MFMAs of the workgroup), one barrier per stage.
//   * B (weights) never touches LDS: wave w is the only consumer of its 32 rows of w, lane (n, k half) loads 16 bytes
//     = 4 consecutive k straight from the (N, K) matrix (L2-resident), three 8-k chunks ahead.
//   * K order inside an 8-k chunk: lane half h supplies k = 8 kb + 4 h + s to MFMA s on BOTH operands.
//   * epilogue straight from the accumulators (C/D layout: column = lane & 31, so one instruction stores two full
//     128-byte row segments); the residual is requested three tiles ahead into the registers the K loop freed.  With
//     two workgroups per CU the other workgroup's MFMAs cover it.
#include <atomic>
#include <type_traits>

#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace pwc {
constexpr int WT = 9;                    // 32-row tiles per wave
constexpr int ROWS = 32 * WT;            // rows per workgroup (288)
constexpr int COLS = 128;                // columns per workgroup (4 waves x 32)
constexpr int KS = 32;                   // k per stage
constexpr int TILE_B = 32 * KS * 4;      // bytes of one 32-row tile of a stage (4096)
constexpr int STAGE_B = WT * TILE_B;     // bytes of a stage (36864)
constexpr unsigned OOB = 0x80000000u;
template <int V>
using IC = std::integral_constant<int, V>;
}  // namespace pwc

struct PwParams {
    const float *x, *w, *bias, *residual;
    float *y;
    int M, N, K, relu;
    int rgroups, cblocks;
    unsigned wbytes, ybytes;
    unsigned long long *trace;  // -DPW_TRACE builds: 16 stamps per wave
};

template <bool RES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void pw_conv_kernel(const PwParams p)
{
    using namespace pwc;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nstages = p.K / KS;
#ifdef PW_TRACE
    unsigned long long ts[16];
    int nts = 0;
#define PW_STAMP() do { if (nts < 12) ts[nts++] = __builtin_readcyclecounter(); } while (0)
    ts[12] = __builtin_amdgcn_s_memrealtime();
#else
#define PW_STAMP() do { } while (0)
#endif
    PW_STAMP();  // 0: start

    // XCD x (workgroup b runs on XCD b % 8: speed only) takes row groups x, x + 8, ...; the column blocks of one row
    // group run back to back on it, so the activation rows are fetched from HBM once
    const int xcd = blockIdx.x & 7, l_ = blockIdx.x >> 3;
    const int cb = l_ % p.cblocks;
    const int rg = (l_ / p.cblocks) * 8 + xcd;
    if (rg >= p.rgroups) return;  // block-uniform
    const int r0 = rg * ROWS, n0 = cb * COLS;

    // ---- A producer: thread = (row prow of each 32-row tile, 16-byte piece pslot of the row's 128 bytes)
    const int prow = tid >> 3, pslot = tid & 7;
    const unsigned avoff = (unsigned)prow * (unsigned)p.K * 4u + (unsigned)pslot * 16u;
    const unsigned awoff = (unsigned)prow * 128u + (unsigned)((pslot ^ ((prow >> 1) & 7)) << 4);
    float4 stg[WT];
    // tile j of stage `stage` through a descriptor over exactly the tile's rows inside [0, M): rows past M and stages
    // past the last one read zeros without traffic (scalar arithmetic only; hipcc clamps in the vector ALU unless the
    // result is forced into a scalar register)
    int nrec[WT];
#pragma unroll
    for (int j = 0; j < WT; ++j) {
        int rows = p.M - r0 - 32 * j;
        rows = rows < 0 ? 0 : rows > 32 ? 32 : rows;
        nrec[j] = __builtin_amdgcn_readfirstlane(rows * p.K * 4);
    }
    auto load_a = [&](int stage, int j) __attribute__((always_inline)) {
        const bool live = stage < nstages;
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(p.x) + (size_t)(r0 + 32 * j) * p.K, 0, live ? nrec[j] : 0, 0x00020000);
        stg[j] = __builtin_bit_cast(
            float4, __builtin_amdgcn_raw_buffer_load_b128(rr, avoff, (live ? stage : 0) * (KS * 4), 0));
    };
    auto store_a = [&](int buf, int j) __attribute__((always_inline)) {
        *reinterpret_cast<float4 *>(lds + buf * STAGE_B + j * TILE_B + awoff) = stg[j];
    };

    // ---- B fragments: lane = (n = lane & 31, k half = lane >> 5), 16 bytes = k 8 kb + 4 h .. + 3 of row n of w
    const int ncol = n0 + 32 * wave + (lane & 31);
    const bool wave_live = n0 + 32 * wave < p.N;  // wave-uniform (N a multiple of 32)
    const unsigned bvoff = wave_live ? (unsigned)ncol * (unsigned)p.K * 4u + (unsigned)(lane >> 5) * 16u : OOB;
    float4 fb[4];
    auto load_b = [&](int stage, int kb) __attribute__((always_inline)) {
        const bool live = stage < nstages;
        const __amdgpu_buffer_rsrc_t rr =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, live ? (int)p.wbytes : 0, 0x00020000);
        fb[kb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                rr, bvoff + 32u * kb, (live ? stage : 0) * (KS * 4), 0));
    };

    // ---- A fragments: lane = (row fr of the tile, k half fh); piece 2 kb + fh of the row, swizzled
    const int fr = lane & 31, fh = lane >> 5, ff = (fr >> 1) & 7;
    unsigned aro[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) aro[b][kb] = (unsigned)(b * STAGE_B + fr * 128 + (((2 * kb + fh) ^ ff) << 4));
    float4 fa[2][3];
    auto read_a = [&](int buf, int kb, int g, int set) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            fa[set][j] = *reinterpret_cast<const float4 *>(lds + aro[buf][kb] + (3 * g + j) * TILE_B);
    };

    // Output / residual element (tile i, accumulator element e) of a lane: row r0 + 32 i + (e & 3) + 8 (e >> 2) +
    // 4 (lane >> 5), column ncol.  Whole tiles (all 32 rows below M) go through a descriptor over the tile with the row
    // of lane half 0 as a scalar offset and ONE constant vector offset -- no vector-ALU work per element; a partial
    // tile (the last one when M is not a multiple of 32) computes and range-checks every offset.
    const unsigned evoff = ((unsigned)(4 * (lane >> 5)) * (unsigned)p.N + (unsigned)ncol) * 4u;
    auto tile_rsrc = [&](const float *base, int rbase, int i) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base) + (size_t)(rbase + 32 * i) * p.N, 0,
                                                 (wave_live && base) ? 32 * p.N * 4 : 0, 0x00020000);
    };
    auto soff_of = [&](int e) __attribute__((always_inline)) { return ((e & 3) + 8 * (e >> 2)) * p.N * 4; };
    auto slow_off = [&](int rbase, int i, int e) __attribute__((always_inline)) {
        const int row = rbase + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        return (row < p.M && wave_live) ? ((unsigned)row * (unsigned)p.N + (unsigned)ncol) * 4u : OOB;
    };

    // ---- prologue: stage 0 into LDS, stage 1 into the staging registers, the B chunks of stage 0; the accumulators
    // start as the residual -- the MFMAs add to it: no epilogue arithmetic, no epilogue loads
#pragma unroll
    for (int j = 0; j < WT; ++j) load_a(0, j);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) load_b(0, kb);
    f32x16 acc[WT];
#pragma unroll
    for (int q = 0; q < WT; ++q) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
        if constexpr (RES) {
            if (r0 + 32 * q + 32 <= p.M) {  // block-uniform
                const __amdgpu_buffer_rsrc_t rr = tile_rsrc(p.residual, r0, q);
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    acc[q][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, evoff, soff_of(e), 0));
            } else {
                const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float *>(p.residual), 0, (int)p.ybytes, 0x00020000);
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    acc[q][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, slow_off(r0, q, e), 0, 0));
            }
        }
    }
    const float bias = (p.bias && wave_live) ? p.bias[ncol] : 0.f;
#pragma unroll
    for (int j = 0; j < WT; ++j) store_a(0, j);
#pragma unroll
    for (int j = 0; j < WT; ++j) load_a(1, j);
    __syncthreads();
    PW_STAMP();  // 1: prologue done
    // + bias as one more k: A = (1, 0) over the lane halves, B = bias of the lane's column
    {
        const float one = lane < 32 ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < WT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(one, bias, acc[q], 0, 0, 0);
    }

    // One stage of this wave = 12 units (kb = chunk of 8 k, g = group of three row tiles) of 12 MFMAs: the unit's three
    // tiles are three independent accumulators taken round-robin, the chunk's four k outermost.  The fragments of the
    // next unit are read in the unit's first slot; units 0-8 also carry the staging duty of tile u -- registers -> LDS
    // (the stage after this one), then the request of the stage after that into the same registers; the B chunk of the
    // next stage is requested when its register set is last used.  Order pinned (sched_barrier after every slot).
    auto stage_body = [&](int s, auto buf_c) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_c)::value;
        read_a(buf, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int kb = u / 3, g = u % 3, set = u & 1;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int m = 3 * k + j;
                    const float av = k == 0 ? fa[set][j].x : k == 1 ? fa[set][j].y : k == 2 ? fa[set][j].z : fa[set][j].w;
                    const float bv = k == 0 ? fb[kb].x : k == 1 ? fb[kb].y : k == 2 ? fb[kb].z : fb[kb].w;
                    acc[3 * g + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[3 * g + j], 0, 0, 0);
                    if (m == 0 && u < 11) read_a(buf, (u + 1) / 3, (u + 1) % 3, set ^ 1);
                    if (u < WT) {
                        if (m == 4) store_a(buf ^ 1, u);
                        if (m == 6) load_a(s + 2, u);
                    }
                    if (g == 2 && m == 11) load_b(s + 1, kb);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    };

    for (int s = 0; s < nstages; s += 2) {
        stage_body(s, IC<0>{});
        __syncthreads();
        stage_body(s + 1, IC<1>{});
        __syncthreads();
#ifdef PW_TRACE
        if (s == 0 || s == 2) PW_STAMP();  // 2, 3: two and four stages done
#endif
    }
    PW_STAMP();  // 4 (2 when there are two stages): K loop done

    // (the 16-pass MFMA needs 18 wait states before its result is read; explicit as in conv_mfma.hip.  Plain vector
    // registers: with an "a" constraint hipcc splits the 256-register budget 128 + 128 and spills)
    int r0e = r0;  // (and the epilogue's descriptors are not built -- and spilled -- before the K loop)
    asm volatile("s_nop 15\n\ts_nop 7"
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                   "+v"(acc[7]), "+v"(acc[8]), "+s"(r0e));

    // ---- epilogue: ReLU and one store per element (two full 128-byte row segments per instruction)
#pragma unroll
    for (int i = 0; i < WT; ++i) {
        if (r0e + 32 * i + 32 <= p.M) {  // block-uniform
            const __amdgpu_buffer_rsrc_t ry = tile_rsrc(p.y, r0e, i);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[i][e];
                if (p.relu) v = fmaxf(v, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, evoff, soff_of(e), 0);
            }
        } else {
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.ybytes, 0x00020000);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[i][e];
                if (p.relu) v = fmaxf(v, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, slow_off(r0e, i, e), 0, 0);
            }
        }
    }
    PW_STAMP();  // stores issued
#ifdef PW_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PW_STAMP();  // stores acknowledged
    if (p.trace && lane == 0) {
        ts[13] = __builtin_amdgcn_s_memrealtime();
        ts[14] = __builtin_amdgcn_s_getreg(63492);
        ts[15] = __builtin_amdgcn_s_getreg(63508);
        unsigned long long *dst = p.trace + ((size_t)blockIdx.x * 4 + wave) * 16;
        for (int i = 0; i < 16; ++i) dst[i] = i < 12 ? (i < nts ? ts[i] : 0) : ts[i];
    }
#endif
}

std::atomic<int> g_pw_override{-1};
unsigned long long *g_pw_trace = nullptr;

}  // namespace

extern "C" void mpsr_debug_set_conv_pointwise(int mode) { g_pw_override = mode; }
extern "C" void mpsr_debug_set_pointwise_trace(void *buf) { g_pw_trace = static_cast<unsigned long long *>(buf); }

namespace mpsr {

int pointwise_override() { return g_pw_override.load(); }

// Shapes the kernel takes: K and N multiples of 32 (an odd stage count runs one stage of zeros), 32-bit byte offsets.
bool pointwise_applies(long long M, int K, int N)
{
    return M > 0 && K >= 32 && K % 32 == 0 && N >= 32 && N % 32 == 0 && M * K * 4 < 0x7f000000LL &&
           M * N * 4 < 0xfffffff0LL && (long long)N * K * 4 < 0x7f000000LL;
}

int conv1x1_pointwise(const float *x, long long M, int K, const float *w, const float *bias, const float *residual,
                      int relu, float *y, int N, hipStream_t s)
{
    using namespace pwc;
    MPSR_REQUIRE(pointwise_applies(M, K, N), "conv1x1_pointwise: unsupported shape (M=%lld K=%d N=%d)", M, K, N);
    PwParams p;
    p.x = x; p.w = w; p.bias = bias; p.residual = residual; p.y = y;
    p.M = (int)M; p.N = N; p.K = K; p.relu = relu;
    p.rgroups = (int)((M + ROWS - 1) / ROWS);
    p.cblocks = ceil_div(N, COLS);
    p.wbytes = (unsigned)((long long)N * K * 4);
    p.ybytes = (unsigned)(M * N * 4);
    p.trace = g_pw_trace;
    const unsigned grid = (unsigned)(ceil_div(p.rgroups, 8) * 8 * p.cblocks);
    const size_t lds_bytes = (size_t)2 * STAGE_B;
    const void *kfn = residual ? reinterpret_cast<const void *>(pw_conv_kernel<true>)
                               : reinterpret_cast<const void *>(pw_conv_kernel<false>);
    MPSR_CHECK_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    if (residual) hipLaunchKernelGGL(pw_conv_kernel<true>, dim3(grid), dim3(256), lds_bytes, s, p);
    else hipLaunchKernelGGL(pw_conv_kernel<false>, dim3(grid), dim3(256), lds_bytes, s, p);
    MPSR_CHECK_LAUNCH("pw_conv_kernel");
    return MPSR_OK;
}

}  // namespace mpsr
