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

// ------------------------------------------------------------------------------------------------------------------
// Streaming form for the short-K, wide-N layers (block3's conv3: K = 256, N = 1024, + residual): per output element
// they move 8 bytes (residual in, result out) for 2 K = 512 flops -- 300 of the layer's 340 MB -- and a kernel that
// loads / stores those in a prologue / epilogue has every CU doing so at once, in bursts, next to an idle matrix pipe.
// Here a PERSISTENT workgroup walks a sequence of 96 x 128 tiles (wave = 3 row tiles x 32 columns, 48 accumulator
// registers) and a tile's traffic rides on its neighbours' K loops:
//   * the residual of tile t + 1 is fetched during the K loop of tile t (16-byte loads, two per thread and stage) into
//     a 48 KB LDS copy of the tile; tile t + 1 starts its accumulators from that copy (48 ds_read_b32, then + bias as
//     one more MFMA k): no residual load ever waits in front of an MFMA, none occupies a register across a K loop;
//   * a finished tile is stored straight from the accumulators (ReLU + one store per element: 2 x 128 contiguous bytes
//     per instruction) and the next tile's MFMAs follow the last store instruction -- the data drains while they run;
//   * the A stages and B chunks run on across tile boundaries as one stream (the next tile's first two stages are
//     requested during this tile's last two).
//   * tile sequence of workgroup b: b, b + G, ...; G is a multiple of 8 x column blocks, so b keeps its XCD and its
//     column block (the same 32 rows of w per wave for its whole life); consecutive workgroups of an XCD share the
//     row group (A from L2).  Tiles past the last row group are all-zero work behind zero-length descriptors.
//   * every descriptor is built in the scalar ALU (base = the row, length = bytes up to row M): rows past M vanish in
//     the range check, no vector-ALU address work.
namespace pws {
constexpr int WT = 3, ROWS = 96, COLS = 128, KS = 32;
constexpr int TILE_B = 32 * KS * 4;          // 4096
constexpr int STAGE_B = WT * TILE_B;         // 12288
constexpr int RES_OFF = 2 * STAGE_B;         // the residual tile behind the two A stages
constexpr int RES_B = ROWS * COLS * 4;       // 49152
constexpr int LDS_B = RES_OFF + RES_B;       // 73728: two workgroups per CU
constexpr unsigned OOB = 0x80000000u;
template <int V>
using IC = std::integral_constant<int, V>;
}  // namespace pws

struct PwsParams {
    const float *x, *w, *bias, *residual;
    float *y;
    int M, N, K, relu;
    int rgroups, cblocks, ntiles;  // ntiles: padded to 8 x column blocks
    unsigned wbytes;
    unsigned long long *trace;  // -DPWS_TRACE builds: 16 stamps per wave (the workgroup's third tile)
};

template <bool RES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void pw_stream_kernel(const PwsParams p)
{
    using namespace pws;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nst = p.K / KS;  // even, >= 4

    // this workgroup's tiles: i -> row group; the column block is fixed
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, l0 = blockIdx.x >> 3;
    const int cb = l0 % p.cblocks;
    const int rgl0 = l0 / p.cblocks, rgstep = (G >> 3) / p.cblocks;
    const int n_my = (p.ntiles - (int)blockIdx.x + G - 1) / G;
    auto r0_of = [&](int i) __attribute__((always_inline)) { return ((rgl0 + i * rgstep) * 8 + xcd) * ROWS; };
    const int n0 = cb * COLS;

    // (lo, hi, left) = address of a row of a tensor and the bytes from there to the tensor's end (row M)
    struct RowRef {
        unsigned lo, hi, left;
    };
    auto row_ref = [&](const float *base, int width, int row, bool live) __attribute__((always_inline)) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(base + (size_t)row * width);
        RowRef r;
        r.lo = (unsigned)a;
        r.hi = (unsigned)(a >> 32);
        r.left = (live && base && row < p.M) ? (unsigned)(p.M - row) * (unsigned)width * 4u : 0u;
        return r;
    };
    // descriptor `rows` rows further on.  max(left - off, 0) in the scalar ALU (hipcc picks the vector ALU's saturating
    // subtract and a waterfall loop otherwise); the asm also keeps hipcc from building a stage's descriptors ahead of
    // its first slot and spilling them.
    auto rsrc_at = [&](RowRef r, int rows, int width) __attribute__((always_inline)) {
        const unsigned off = (unsigned)rows * (unsigned)width * 4u;
        unsigned left;
        asm volatile("s_sub_u32 %0, %1, %2\n\ts_cselect_b32 %0, 0, %0" : "=&s"(left) : "s"(r.left), "s"(off) : "scc");
        asm volatile("" : "+s"(r.lo), "+s"(r.hi));
        const unsigned long long a = (((unsigned long long)r.hi << 32) | r.lo) + off;
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(a), 0, (int)left, 0x00020000);
    };

    // ---- A producer: thread = (row prow of each 32-row tile, 16-byte piece pslot of the row's 128 bytes)
    const int prow = tid >> 3, pslot = tid & 7;
    const unsigned avoff = (unsigned)prow * (unsigned)p.K * 4u + (unsigned)pslot * 16u;
    const unsigned awoff = (unsigned)prow * 128u + (unsigned)((pslot ^ ((prow >> 1) & 7)) << 4);
    float4 stg[WT];
    auto load_a = [&](RowRef xr, int stage, int j) __attribute__((always_inline)) {
        stg[j] = __builtin_bit_cast(
            float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_at(xr, 32 * j, p.K), avoff, stage * (KS * 4), 0));
    };
    auto store_a = [&](int buf, int j) __attribute__((always_inline)) {
        *reinterpret_cast<float4 *>(lds + buf * STAGE_B + j * TILE_B + awoff) = stg[j];
    };

    // ---- B fragments: lane = (n = lane & 31, k half = lane >> 5), 16 bytes = k 8 kb + 4 h .. + 3 of row n of w
    const int ncol = n0 + 32 * wave + (lane & 31);
    const bool wave_live = n0 + 32 * wave < p.N;  // wave-uniform
    const unsigned bvoff = wave_live ? (unsigned)ncol * (unsigned)p.K * 4u + (unsigned)(lane >> 5) * 16u : OOB;
    float4 fb[4];
    auto load_b = [&](bool live, int stage, int kb) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rr =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, live ? (int)p.wbytes : 0, 0x00020000);
        fb[kb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, bvoff + 32u * kb, stage * (KS * 4), 0));
    };

    // ---- A fragments: lane = (row fr of the tile, k half fh); piece 2 kb + fh of the row, swizzled
    const int fr = lane & 31, fh = lane >> 5, ff = (fr >> 1) & 7;
    unsigned aro[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) aro[kb] = (unsigned)(fr * 128 + (((2 * kb + fh) ^ ff) << 4));
    float4 fa[2][3];
    auto read_a = [&](int buf, int kb, int set) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            fa[set][j] = *reinterpret_cast<const float4 *>(lds + buf * STAGE_B + j * TILE_B + aro[kb]);
    };

    // ---- residual staging: piece P = tid + 256 jj of the 96 x 128 tile (row-major, 32 pieces per row): row
    // (tid >> 5) + 8 jj, columns 4 (tid & 31) .. + 3
    const unsigned rvoff = ((unsigned)(tid >> 5) * (unsigned)p.N + (unsigned)(n0 + 4 * (tid & 31))) * 4u;
    const unsigned rwoff = (unsigned)(RES_OFF + (tid >> 5) * 512 + (tid & 31) * 16);
    float4 rst[6];
    auto load_r = [&](RowRef rr, int jj) __attribute__((always_inline)) {
        rst[jj % 6] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_at(rr, 8 * jj, p.N), rvoff, 0, 0));
    };
    auto store_r = [&](int jj) __attribute__((always_inline)) {
        *reinterpret_cast<float4 *>(lds + rwoff + jj * 4096) = rst[jj % 6];
    };
    // element (q, e) of a lane: row 32 q + (e & 3) + 8 (e >> 2) + 4 (lane >> 5) of the tile, column 32 wave + (lane & 31)
    const unsigned rroff = (unsigned)(RES_OFF + (4 * (lane >> 5)) * 512 + (32 * wave + (lane & 31)) * 4);
    const unsigned evoff = ((unsigned)(4 * (lane >> 5)) * (unsigned)p.N + (unsigned)ncol) * 4u;

    const float bias = (p.bias && wave_live) ? p.bias[ncol] : 0.f;
    const float one = lane < 32 ? 1.f : 0.f;
    const float relu_lo = p.relu ? 0.f : -__builtin_inff();
    f32x16 acc[WT];

    // One stage = 4 units (kb) of 12 MFMAs on the three accumulators, k outermost.  Slot D = 12 kb + m carries: the next
    // unit's fragment reads (m = 0); the A staging of tile j (registers -> LDS at D = 4 + 12 j, the request two stages on
    // at D + 2); the B chunk of the next stage (m = 11); and the residual duty -- DUTY 1 / 2: the first / second six
    // pieces of the next tile's residual requested (D = 7 + 6 n); DUTY 2 / 3: the first / second six written to LDS
    // (D = 9 + 6 n).
    auto stage_body = [&](auto buf_c, auto duty_c, int s, RowRef xc, bool clive, RowRef xn, bool nlive, RowRef rn)
                          __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_c)::value, DUTY = decltype(duty_c)::value;
        read_a(buf, 0, 0);
        // stage s + 2 of this tile, or stage s + 2 - nst of the next one (scalar selects)
        const bool wrap2 = s + 2 >= nst;
        RowRef xa;
        xa.lo = wrap2 ? xn.lo : xc.lo;
        xa.hi = wrap2 ? xn.hi : xc.hi;
        xa.left = wrap2 ? xn.left : xc.left;
        const int a_st = wrap2 ? s + 2 - nst : s + 2;
        const bool wrap1 = s + 1 >= nst;
        const int b_st = wrap1 ? 0 : s + 1;
        const bool b_live = wrap1 ? nlive : clive;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int set = kb & 1;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int m = 3 * k + j, D = 12 * kb + m;
                    const float av = k == 0 ? fa[set][j].x : k == 1 ? fa[set][j].y : k == 2 ? fa[set][j].z : fa[set][j].w;
                    const float bv = k == 0 ? fb[kb].x : k == 1 ? fb[kb].y : k == 2 ? fb[kb].z : fb[kb].w;
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
                    if (m == 0 && kb < 3) read_a(buf, kb + 1, set ^ 1);
                    if (D >= 4 && D < 4 + 12 * WT && (D - 4) % 12 == 0) store_a(buf ^ 1, (D - 4) / 12);
                    if (D >= 6 && D < 6 + 12 * WT && (D - 6) % 12 == 0) load_a(xa, a_st, (D - 6) / 12);
                    if (m == 11) load_b(b_live, b_st, kb);
                    if constexpr (RES) {
                        if ((DUTY == 1 || DUTY == 2) && D >= 7 && D < 7 + 36 && (D - 7) % 6 == 0)
                            load_r(rn, (DUTY - 1) * 6 + (D - 7) / 6);
                        if ((DUTY == 2 || DUTY == 3) && D >= 9 && D < 9 + 36 && (D - 9) % 6 == 0)
                            store_r((DUTY - 2) * 6 + (D - 9) / 6);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    };

    // ---- prologue: the first tile's residual into its LDS copy, its stage 0 into LDS, stage 1 into the staging
    // registers, the B chunks of stage 0
    {
        const RowRef x0 = row_ref(p.x, p.K, r0_of(0), true);
#pragma unroll
        for (int j = 0; j < WT; ++j) load_a(x0, 0, j);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) load_b(true, 0, kb);
        if constexpr (RES) {
            const RowRef r0r = row_ref(p.residual, p.N, r0_of(0), true);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int jj = 0; jj < 6; ++jj) load_r(r0r, 6 * h + jj);
#pragma unroll
                for (int jj = 0; jj < 6; ++jj) store_r(6 * h + jj);
            }
        }
#pragma unroll
        for (int j = 0; j < WT; ++j) store_a(0, j);
#pragma unroll
        for (int j = 0; j < WT; ++j) load_a(x0, 1, j);
        __syncthreads();
    }

#ifdef PWS_TRACE
    unsigned long long ts[16];
    int nts = 0;
    for (int k = 0; k < 16; ++k) ts[k] = 0;
#define PWS_STAMP(cond) do { if ((cond) && nts < 16) ts[nts++] = __builtin_readcyclecounter(); } while (0)
#else
#define PWS_STAMP(cond) do { } while (0)
#endif
    for (int i = 0; i < n_my; ++i) {
        const int r0c = r0_of(i), r0n = r0_of(i + 1);
        const bool nlive = i + 1 < n_my;
        PWS_STAMP(i == 2);  // 0: tile start
        const RowRef xc = row_ref(p.x, p.K, r0c, true), xn = row_ref(p.x, p.K, r0n, nlive);
        const RowRef rn = row_ref(p.residual, p.N, r0n, nlive);
        // the accumulators start from the residual (its LDS copy) ...
#pragma unroll
        for (int q = 0; q < WT; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                acc[q][e] = RES ? *reinterpret_cast<const float *>(lds + rroff + (32 * q + (e & 3) + 8 * (e >> 2)) * 512) : 0.f;
        // ... + bias as one more k: A = (1, 0) over the lane halves, B = bias of the lane's column
#pragma unroll
        for (int q = 0; q < WT; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(one, bias, acc[q], 0, 0, 0);
        PWS_STAMP(i == 2);  // 1: accumulators initialised
        stage_body(IC<0>{}, IC<0>{}, 0, xc, true, xn, nlive, rn);
        PWS_STAMP(i == 2);  // 2: stage 0 issued
        __syncthreads();  // (every wave has read its part of the residual copy: the next tile's may be written)
        PWS_STAMP(i == 2);  // 3: barrier
        stage_body(IC<1>{}, IC<1>{}, 1, xc, true, xn, nlive, rn);
        PWS_STAMP(i == 2);  // 4
        __syncthreads();
        PWS_STAMP(i == 2);  // 5
        stage_body(IC<0>{}, IC<2>{}, 2, xc, true, xn, nlive, rn);
        PWS_STAMP(i == 2);  // 6
        __syncthreads();
        PWS_STAMP(i == 2);  // 7
        stage_body(IC<1>{}, IC<3>{}, 3, xc, true, xn, nlive, rn);
        PWS_STAMP(i == 2);  // 8
        __syncthreads();
        PWS_STAMP(i == 2);  // 9
        for (int s = 4; s < nst; s += 2) {
            stage_body(IC<0>{}, IC<0>{}, s, xc, true, xn, nlive, rn);
            PWS_STAMP(i == 2 && s == 4);  // 10
            __syncthreads();
            PWS_STAMP(i == 2 && s == 4);  // 11
            stage_body(IC<1>{}, IC<0>{}, s + 1, xc, true, xn, nlive, rn);
            __syncthreads();
        }
        PWS_STAMP(i == 2);  // 12: K loop done
        // ---- the tile's stores (the 16-pass MFMA needs 18 wait states before its result is read; explicit as in
        // conv_mfma.hip): ReLU + one store per element, two full 128-byte row segments per instruction
        int r0s = r0c;
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+s"(r0s));
        if (r0s + ROWS <= p.M) {  // block-uniform: the whole tile is inside the tensor -- one descriptor, the row of
                                  // lane half 0 as the scalar offset
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
                p.y + (size_t)r0s * p.N, 0, wave_live ? ROWS * p.N * 4 : 0, 0x00020000);
#pragma unroll
            for (int q = 0; q < WT; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(acc[q][e], relu_lo)), ry, evoff,
                                                          (32 * q + (e & 3) + 8 * (e >> 2)) * p.N * 4, 0);
        } else {
            const RowRef yc = row_ref(p.y, p.N, r0s, wave_live);
#pragma unroll
            for (int q = 0; q < WT; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(acc[q][e], relu_lo)),
                                                          rsrc_at(yc, 32 * q + (e & 3) + 8 * (e >> 2), p.N), evoff, 0, 0);
        }
        PWS_STAMP(i == 2);  // 13: stores issued
    }
#ifdef PWS_TRACE
    if (p.trace && lane == 0) {
        unsigned long long *dst = p.trace + ((size_t)blockIdx.x * 4 + wave) * 16;
        for (int k = 0; k < 16; ++k) dst[k] = ts[k];
    }
#endif
}

std::atomic<int> g_pw_override{-1};  // -1 heuristic, 0 never, 1 the 288 x 128 kernel, 2 the streaming kernel
std::atomic<int> g_pws_per_cu{2};
unsigned long long *g_pw_trace = nullptr;

}  // namespace

extern "C" void mpsr_debug_set_conv_pointwise(int mode) { g_pw_override = mode; }
extern "C" void mpsr_debug_set_pointwise_stream_per_cu(int n) { g_pws_per_cu = n > 0 ? n : 2; }
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


// The streaming form: K a multiple of 64 and >= 128 (whole pairs of 32-k stages, four of them carry the residual
// duties), N a multiple of 32, few enough column blocks for a persistent grid whose stride keeps a workgroup on one
// column block.
bool pointwise_stream_applies(long long M, int K, int N)
{
    return pointwise_applies(M, K, N) && K % 64 == 0 && K >= 128 && (N + 127) / 128 <= 32;
}

int conv1x1_pointwise_stream(const float *x, long long M, int K, const float *w, const float *bias,
                             const float *residual, int relu, float *y, int N, hipStream_t s)
{
    using namespace pws;
    MPSR_REQUIRE(pointwise_stream_applies(M, K, N), "conv1x1_pointwise_stream: unsupported shape (M=%lld K=%d N=%d)", M, K, N);
    PwsParams p;
    p.x = x; p.w = w; p.bias = bias; p.residual = residual; p.y = y;
    p.M = (int)M; p.N = N; p.K = K; p.relu = relu;
    p.rgroups = (int)((M + ROWS - 1) / ROWS);
    p.cblocks = ceil_div(N, COLS);
    p.ntiles = ceil_div(p.rgroups, 8) * 8 * p.cblocks;
    p.wbytes = (unsigned)((long long)N * K * 4);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 256;
        MPSR_CHECK_HIP(hipGetDevice(&dev));
        MPSR_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        cus = n;
    }
    // persistent grid: g_pws_per_cu workgroups per CU, rounded down to a multiple of 8 x column blocks
    const int unit = 8 * p.cblocks;
    int grid = g_pws_per_cu.load() * cus / unit * unit;
    if (grid < unit) grid = unit;
    if (grid > p.ntiles) grid = p.ntiles;
    p.trace = g_pw_trace;
    const size_t lds_bytes = (size_t)LDS_B;
    const void *kfn = residual ? reinterpret_cast<const void *>(pw_stream_kernel<true>)
                               : reinterpret_cast<const void *>(pw_stream_kernel<false>);
    MPSR_CHECK_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    if (residual) hipLaunchKernelGGL(pw_stream_kernel<true>, dim3(grid), dim3(256), lds_bytes, s, p);
    else hipLaunchKernelGGL(pw_stream_kernel<false>, dim3(grid), dim3(256), lds_bytes, s, p);
    MPSR_CHECK_LAUNCH("pw_stream_kernel");
    return MPSR_OK;
}

}  // namespace mpsr
