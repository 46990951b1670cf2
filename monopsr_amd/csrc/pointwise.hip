// 1x1 convolution (pointwise GEMM) of the trunk's wide layers on fp32 MFMA:
//     y[m][n] = act( sum_k x[m][k] w[n][k] + bias[n] + residual[m][n] ),   m = pixel (NHWC row), K = C.
// ResNet-101 block3's conv1 / conv3, the projection shortcut, block2's conv3 and the squash layers (reference graph
// object_detection/nets/resnet_v1.py:104-131 bottleneck(), monopsr/core/feature_extractors/resnet.py) -- 9 of the
// step's 16 ms went through the general implicit GEMM (conv_mfma.hip), whose tiles pass BOTH operands through LDS
// (one ds_write_b128 per 16-byte piece, 20-45 cycles of matrix-pipe time each: profiles/r03_mfma_with_lds_stores.txt),
// re-read them as 4 fragments per 12 MFMAs, and load the residual / store the result in an epilogue.  The short-K,
// wide-N layers (block3's conv3: K = 256, N = 1024, + residual) move 8 bytes per output element (residual in, result
// out) for 2 K = 512 flops -- 300 of the layer's 340 MB -- and with prologue / epilogue traffic every CU loads and stores
// at once, in bursts, next to an idle matrix pipe.
//
// Here a PERSISTENT workgroup (4 waves) walks a sequence of 96 x 128 tiles (wave = 3 row tiles x 32 columns, 48
// accumulator registers per set) and a tile's traffic rides on its neighbours' K loops:
//   * A (activations) is shared by the four waves through LDS: 32 k per stage, two stages (128-byte rows, the 16-byte
//     pieces XOR-swizzled by (row >> 1) & 7: ds_read_b128 fragment reads and ds_write_b128 fills are conflict-free);
//     global -> registers a whole stage ahead, registers -> LDS early in the next stage, one barrier per stage.
//   * B (weights) never touches LDS: wave w is the only consumer of its 32 rows of w, lane (n, k half) loads 16 bytes
//     = 4 consecutive k straight from the (N, K) matrix (L2-resident), three 8-k chunks ahead.
//   * K order inside an 8-k chunk: lane half h supplies k = 8 kb + 4 h + s to MFMA s on BOTH operands.
//   * the residual of tile t + 1 is fetched during the K loop of tile t (16-byte loads, six per thread in each of two
//     stages) into a 48 KB LDS copy of the tile; tile t + 1 starts its accumulators from that copy (48 ds_read_b32,
//     then + bias as one more MFMA k: A = 1, B = bias): no residual load waits in front of an MFMA, none occupies a
//     register across a K loop, no vector-ALU arithmetic in front of the stores except the ReLU;
//   * two accumulator sets alternate: a finished tile is stored from its set during the first eight stages of the next
//     tile (ReLU + one store per element -- 2 x 128 contiguous bytes per instruction -- every eighth MFMA slot; issued
//     in one burst behind the K loop the same stores held the wave for 9000 cycles);
//   * the A stages and B chunks run on across tile boundaries as one stream (the next tile's first two stages are
//     requested during this tile's last two);
//   * tile sequence of workgroup b: b, b + G, ...; G is a multiple of 8 x column blocks, so b keeps its XCD and its
//     column block (the same 32 rows of w per wave for its whole life); consecutive workgroups of an XCD share the
//     row group (A from L2).  Tiles past the last row group are all-zero work behind zero-length descriptors;
//   * every descriptor is built in the scalar ALU (base = the row, length = bytes up to row M): rows past M vanish in
//     the range check, no vector-ALU address work.
// M = batch x 144 pixels on the 12x12 trunk maps: 96-row groups divide it exactly, and at batch 256 the 384 groups x
// N / 128 column blocks are 1.5 (N = 256), 3 (512) or 6 (1024) tiles for each of the 512 resident workgroups.
// What is left (tools/pws_trace.py): one vmcnt counts loads AND stores in order, so the first wait after a store or a
// residual request also waits for those -- stages 1-2 of a tile take ~3x a plain stage.
#include <atomic>
#include <type_traits>

#include "common.h"
#include "wino3_filter.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// PWS_WIDE (A/B builds only, default 0): the MFMA operands swapped (the matrix pipe computes the TRANSPOSED 32 x 32 block,
// W X^T), so that a lane's four consecutive accumulator registers are four consecutive output CHANNELS of one pixel
// row: a tile leaves as 12 buffer_store_dwordx4 per lane instead of 48 buffer_store_dword and starts from the residual
// copy with 12 ds_read_b128 instead of 48 ds_read_b32.  Same products, same K order: bit-identical results (tested).
// Measured SLOWER (r04, whole step, interleaved on one board: 15.20 -> 15.47 ms; 15.42 with the stores spread thinner,
// profiles/r04_pointwise_wide_stores.txt): a dword store writes 2 rows x 128 contiguous bytes = two full lines per
// instruction, the 16-byte form 32 rows x 32 bytes = 32 partial lines -- four times the write requests for the same
// bytes.  The guide's "widen the epilogue stores" applies to row-per-lane layouts, not to stores that are already
// full lines.
#ifndef PWS_WIDE
#define PWS_WIDE 0
#endif
// Cache-policy bits of the residual loads and of the result stores: 2 = non-temporal.  The residual passes through ONCE
// (151 MB on block3's conv3) next to operands that are re-read for the whole launch -- the 1 MB of filters by every
// tile, a row group's activations by all its column blocks: marked streaming it stops evicting those from the 4 MB L2.
// Step, interleaved on one board, three boards: 13.50 -> 13.41 ms (-1.2 %) with the residual loads alone.  The result
// stores are a smaller and mixed effect: streaming them helps the launches WITHOUT a residual (conv1 / shortcut / tap
// GEMMs: 13.41 -> 13.39) and costs on conv3 (its result is the next launch's input: 13.43 with every store streaming),
// hence the default below.  The same hint on the F(4x4,3x3) / sixteen-product kernels' stores (W4_NT, W3Z_STORE_AUX)
// measured +0.08 / +0.05 ms, on their transformed-filter loads +0.05 / +0.09 (W4_BAUX, W3Z_BAUX), on F(4x4,3x3)'s patch
// loads +0.13 (W4_AAUX), on the upsampling gather's loads and stores (UPC_NT) equal: everything that is read twice or
// read back at once wants the default policy; they stay off.
#ifndef PWS_STORE_AUX
#define PWS_STORE_AUX (RES ? 0 : 2)
#endif
#ifndef PWS_A_AUX
#define PWS_A_AUX 0  // the activations' loads (A/B builds): streaming them costs +0.2 .. +0.36 ms -- every row group is read
                     // by all of its column blocks and wants to stay in L2
#endif
#ifndef PWS_RES_AUX
#define PWS_RES_AUX 2
#endif
#ifndef PWS_WSTEP
#define PWS_WSTEP 16  // wide form: MFMA slots between two stores of the previous tile (16: three per stage, four stages)
#endif

namespace pws {
constexpr bool WIDE = PWS_WIDE != 0;
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
    // tail job: the F(3x3,3x3) filter transform of the 3x3 layer that follows (wino3_filter.h), shared by all workgroups
    const float *fw;
    float *fu;
    int fN, fC, fform;  // fform 0: F(3x3,3x3) filters (25 positions), 1: the sixteen-product form
    // MASK instantiations (training, data-gradient launches): a bit of mask[(row >> 5) * N + column] says whether the
    // output element is kept or written as zero -- the ReLU mask of the tensor this gradient belongs to, applied in the
    // store path instead of by an elementwise pass over the result.  Bit b of a word <-> row 32 g + (b & 3) +
    // 8 ((b >> 2) & 3) + 4 (b >> 4): the accumulator layout of v_mfma_f32_32x32x2_f32 (lane half h = b >> 4 holds element
    // e = b & 15), so a lane's 16 elements of a 32-row tile are one halfword.
    // EMIT instantiations (training, forward launches): the same words written for THIS launch's result (y > 0), one
    // 2-byte store per lane and 32 x 32 block, so that no pass has to read y back to make them (mpsr_relu_bitmask).
    const unsigned *mask;
    unsigned *emit;
    unsigned maskbytes;
};

// LONG: K >= 256 (eight or more stages: the stores of a tile are spread over the next tile's first eight); otherwise
// four stages carry them.
template <bool RES, bool LONG, bool MASK = false, bool EMIT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void pw_conv_kernel(const PwsParams p)
{
    using namespace pws;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nst = p.K / KS;  // even, >= 4
#ifdef PWS_PRIO  // A/B builds (r06): issue priority between the two co-resident workgroups' waves of a SIMD, by the
    // hardware wave slot (HW_ID.wave_id, bits 3:0): 1 = the odd slot runs at priority 1 for the whole launch; 2 = the
    // two alternate tile by tile.  Measured over three interleaved rounds of the step (tools/step_libs.sh): 13.266 ms
    // without, 13.285 with 1, 13.297 with 2 -- two independent persistent workgroups are not the compute / load pair of
    // one workgroup that a static priority helps (it did help winograd4.hip: W4_PRIO); not compiled in.
    const int slot_bit = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1;
    if (PWS_PRIO == 1 && slot_bit) __builtin_amdgcn_s_setprio(1);
#endif

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
        const bool in = live && base && row >= 0 && row < p.M;  // (a dead reference points at row 0 with length 0)
        const unsigned long long a = reinterpret_cast<unsigned long long>(base + (size_t)(in ? row : 0) * width);
        RowRef r;
        r.lo = (unsigned)a;
        r.hi = (unsigned)(a >> 32);
        r.left = in ? (unsigned)(p.M - row) * (unsigned)width * 4u : 0u;
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
            float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_at(xr, 32 * j, p.K), avoff, stage * (KS * 4), PWS_A_AUX));
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
    // (wide form: the 16-byte pieces of a row are XOR-swizzled by row & 15 -- lanes of one ds_read_b128 group then hit
    // 16 different bank quads although their rows are 512 bytes apart; row & 15 = (tid >> 5) + 8 (jj & 1))
    const unsigned rwoff = WIDE ? (unsigned)(RES_OFF + (tid >> 5) * 512 + (((tid & 31) ^ (tid >> 5)) << 4))
                                : (unsigned)(RES_OFF + (tid >> 5) * 512 + (tid & 31) * 16);
    float4 rst[6];
    auto load_r = [&](RowRef rr, int jj) __attribute__((always_inline)) {
        rst[jj % 6] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_at(rr, 8 * jj, p.N), rvoff, 0, PWS_RES_AUX));
    };
    auto store_r = [&](int jj) __attribute__((always_inline)) {
        *reinterpret_cast<float4 *>(lds + ((WIDE && (jj & 1)) ? (rwoff ^ 128u) : rwoff) + jj * 4096) = rst[jj % 6];
    };
    // element (q, e) of a lane: row 32 q + (e & 3) + 8 (e >> 2) + 4 (lane >> 5) of the tile, column 32 wave + (lane & 31)
    // wide form: element (q, e) of a lane: row 32 q + (lane & 31), column 32 wave + 8 (e >> 2) + 4 (lane >> 5) + (e & 3);
    // a register quad g = e >> 2 is piece 8 wave + 2 g + (lane >> 5) of the row (swizzled: ^ (row & 15) = ^ (lane & 15))
    const unsigned rroff = WIDE ? (unsigned)(RES_OFF + (lane & 31) * 512 + (((8 * wave + (lane >> 5)) ^ (lane & 15)) << 4))
                                : (unsigned)(RES_OFF + (4 * (lane >> 5)) * 512 + (32 * wave + (lane & 31)) * 4);
    const unsigned evoff = WIDE ? ((unsigned)(lane & 31) * (unsigned)p.N + (unsigned)(n0 + 32 * wave + 4 * (lane >> 5))) * 4u
                                : ((unsigned)(4 * (lane >> 5)) * (unsigned)p.N + (unsigned)ncol) * 4u;

    const float bias = (p.bias && wave_live) ? p.bias[ncol] : 0.f;
    const float one = lane < 32 ? 1.f : 0.f;
    const float bias_k0 = lane < 32 ? bias : 0.f;  // wide form: the bias is the A operand (row index = channel), k = 0 only
    const float relu_lo = p.relu ? 0.f : -__builtin_inff();
    // MASK: the three mask words (32 rows x this lane's column) of the tile that accumulates in set S, requested when the
    // tile starts and used a whole tile later; out_bits() = what a store writes for element (q, e) of a set
    static_assert(!(MASK && WIDE), "the mask words follow the narrow accumulator layout");
    unsigned mw[2][WT];
    const unsigned mvoff = wave_live ? (unsigned)ncol * 4u : OOB;
    auto load_mask = [&](auto set_c, int r0, bool live) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value;
        const __amdgpu_buffer_rsrc_t rr =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(p.mask), 0, live ? (int)p.maskbytes : 0, 0x00020000);
#pragma unroll
        for (int q = 0; q < WT; ++q)
            mw[S][q] = __builtin_amdgcn_raw_buffer_load_b32(rr, mvoff, (unsigned)((live ? r0 : 0) / 32 + q) * (unsigned)p.N * 4u, 0);
    };
    const unsigned mshift = 16u * (unsigned)(lane >> 5);
    unsigned mb[2][WT];  // EMIT: the halfword of set S's tile gathering while its elements are stored
    const unsigned emvoff = wave_live ? (unsigned)ncol * 4u + 2u * (unsigned)(lane >> 5) : OOB;
    f32x16 acc[2][WT];  // a tile accumulates in one set while the previous tile is stored from the other
    auto out_bits = [&](auto set_c, int q, int e) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value;
        if constexpr (MASK) {
            const int keep = __builtin_amdgcn_sbfe((int)mw[S][q], mshift + (unsigned)e, 1u);
            const float v = acc[S][q][e];  // (bit_cast of the vector-element lvalue itself picks element 0)
            return (unsigned)keep & __builtin_bit_cast(unsigned, v);
        } else {
            const float v = fmaxf(acc[S][q][e], relu_lo);
            if constexpr (EMIT) mb[S][q] = (e ? mb[S][q] : 0u) | (v > 0.f ? 1u << e : 0u);
            return __builtin_bit_cast(unsigned, v);
        }
    };
    // EMIT: after element 15 of block q of the tile at row r0 the halfword is complete
    auto emit_bits = [&](auto set_c, int q, int r0, bool live) __attribute__((always_inline)) {
        constexpr int S = decltype(set_c)::value;
        const __amdgpu_buffer_rsrc_t rr =
            __builtin_amdgcn_make_buffer_rsrc(p.emit, 0, live ? (int)p.maskbytes : 0, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b16((short)mb[S][q], rr, emvoff, (unsigned)((live ? r0 : 0) / 32 + q) * (unsigned)p.N * 4u, 0);
    };
#pragma unroll
    for (int q = 0; q < WT; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[1][q][e] = 0.f;

    // wide form: register quad idx = 4 q + g of accumulator set SET -> ReLU -> 16 bytes of row 32 q + (lane & 31)
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    auto store_quad = [&](auto set_c, int idx, RowRef yr) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value;
        const int q = idx / 4, g = idx % 4;
        u32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = __builtin_bit_cast(unsigned, fmaxf(acc[SET][q][4 * g + c], relu_lo));
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_at(yr, 32 * q, p.N), evoff, 32 * g, 0);
    };

    // One stage = 4 units (kb) of 12 MFMAs on the three accumulators, k outermost.  Slot D = 12 kb + m carries: the next
    // unit's fragment reads (m = 0); the A staging of tile j (registers -> LDS at D = 4 + 12 j, the request two stages on
    // at D + 2); the B chunk of the next stage (m = 11); and the residual duty -- DUTY 1 / 2: the first / second six
    // pieces of the next tile's residual requested (D = 9 + 6 n); DUTY 2 / 3: the first / second six written to LDS
    // (D = 7 + 6 n).
    // SE0 >= 0: also ReLU + store of elements SE0, SE0 + 1, ... of the previous tile (the other accumulator set), one
    // every SSTEP slots (D = 1 + SSTEP n) -- the 48 stores of a tile are spread over its first eight stages (four when
    // K = 128 / 192): one in-order counter covers loads and stores, so every wait for an A / B request also waits for
    // the stores issued before it, and those complete sooner the thinner they are spread (12 KB per wave in one burst
    // behind the K loop held the wave for 9000 cycles; over three stages the stages after them took 3x as long).
    auto stage_body = [&](auto set_c, auto buf_c, auto duty_c, auto se0_c, auto sstep_c, int s, RowRef xc, bool clive,
                          RowRef xn, bool nlive, RowRef rn, RowRef yp, int r0p = 0, bool plive = false)
                          __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value, buf = decltype(buf_c)::value, DUTY = decltype(duty_c)::value,
                      SE0 = decltype(se0_c)::value, SSTEP = decltype(sstep_c)::value;
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
                    acc[SET][j] = WIDE ? __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[SET][j], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[SET][j], 0, 0, 0);
                    if (m == 0 && kb < 3) read_a(buf, kb + 1, set ^ 1);
#ifndef PWS_SKIP_STORE  // (timing experiments only: tools/README.md)
                    if constexpr (SE0 >= 0 && WIDE) {
                        if (D % SSTEP == 1 && SE0 + D / SSTEP < 4 * WT) store_quad(IC<SET ^ 1>{}, SE0 + D / SSTEP, yp);
                    } else if constexpr (SE0 >= 0) if (D % SSTEP == 1 && SE0 + D / SSTEP < 16 * WT) {
                        const int q = (SE0 + D / SSTEP) / 16, e = (SE0 + D / SSTEP) % 16;
                        __builtin_amdgcn_raw_buffer_store_b32(out_bits(IC<SET ^ 1>{}, q, e),
                                                              rsrc_at(yp, 32 * q + (e & 3) + 8 * (e >> 2), p.N), evoff, 0, PWS_STORE_AUX);
                        if constexpr (EMIT) if (e == 15) emit_bits(IC<SET ^ 1>{}, q, r0p, plive);
                    }
#endif
                    if (D >= 4 && D < 4 + 12 * WT && (D - 4) % 12 == 0) store_a(buf ^ 1, (D - 4) / 12);
                    if (D >= 6 && D < 6 + 12 * WT && (D - 6) % 12 == 0) load_a(xa, a_st, (D - 6) / 12);
                    if (m == 11) load_b(b_live, b_st, kb);
                    if constexpr (RES) {
                        // (stage 2 first writes piece n of the first six out of its register, then requests piece
                        // 6 + n into it)
                        if ((DUTY == 2 || DUTY == 3) && D >= 7 && D < 7 + 36 && (D - 7) % 6 == 0)
                            store_r((DUTY - 2) * 6 + (D - 7) / 6);
#ifndef PWS_SKIP_RES
                        if ((DUTY == 1 || DUTY == 2) && D >= 9 && D < 9 + 36 && (D - 9) % 6 == 0)
                            load_r(rn, (DUTY - 1) * 6 + (D - 9) / 6);
#endif
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
    auto tile_body = [&](auto set_c, int i) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value;
        const int r0c = r0_of(i), r0n = r0_of(i + 1);
        const bool nlive = i + 1 < n_my;
#ifdef PWS_PRIO
        if (PWS_PRIO == 2) {
            if ((SET ^ slot_bit) & 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
        PWS_STAMP(i == 2);  // 0: tile start
        const RowRef xc = row_ref(p.x, p.K, r0c, true), xn = row_ref(p.x, p.K, r0n, nlive);
        const RowRef rn = row_ref(p.residual, p.N, r0n, nlive);
        const RowRef yp = row_ref(p.y, p.N, r0_of(i - 1), i > 0 && wave_live);
        if constexpr (MASK) load_mask(IC<SET>{}, r0c, r0c < p.M);
        // the accumulators start from the residual (its LDS copy) ...
        if constexpr (WIDE) {
#pragma unroll
            for (int q = 0; q < WT; ++q)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 r4 = RES ? *reinterpret_cast<const float4 *>(lds + (rroff ^ (unsigned)(32 * g)) + q * 16384)
                                          : float4{0.f, 0.f, 0.f, 0.f};
                    acc[SET][q][4 * g] = r4.x;
                    acc[SET][q][4 * g + 1] = r4.y;
                    acc[SET][q][4 * g + 2] = r4.z;
                    acc[SET][q][4 * g + 3] = r4.w;
                }
            // ... + bias as one more k: A = bias of the lane's channel row (k = 0 only), B = 1
#pragma unroll
            for (int q = 0; q < WT; ++q) acc[SET][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(bias_k0, 1.f, acc[SET][q], 0, 0, 0);
        } else {
#pragma unroll
            for (int q = 0; q < WT; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    acc[SET][q][e] = RES ? *reinterpret_cast<const float *>(lds + rroff + (32 * q + (e & 3) + 8 * (e >> 2)) * 512) : 0.f;
            // ... + bias as one more k: A = (1, 0) over the lane halves, B = bias of the lane's column
#pragma unroll
            for (int q = 0; q < WT; ++q) acc[SET][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(one, bias, acc[SET][q], 0, 0, 0);
        }
        // stage (s, buffer s & 1, residual duty, first stored element, slots per store)
#define PWS_STAGE(S_, DUTY_, SE0_, SSTEP_)                                                                            \
    stage_body(IC<SET>{}, IC<(S_) & 1>{}, IC<DUTY_>{}, IC<SE0_>{}, IC<SSTEP_>{}, S_, xc, true, xn, nlive, rn, yp,     \
               r0_of(i - 1), i > 0 && r0_of(i - 1) < p.M);                                                            \
    __syncthreads()
        if constexpr (WIDE) {
            // 12 quad stores: PWS_WSTEP slots apart (48 / PWS_WSTEP per stage) over the first stages; the residual duties
            // as before in stages 1-3.  (after stage 0's barrier every wave has read its part of the residual copy:
            // the next tile's may be written)
            constexpr int WS = (!LONG && PWS_WSTEP > 16) ? 16 : PWS_WSTEP;  // (short K: four stages carry the 12 stores)
            constexpr int PS = 48 / WS;                                      // stores per stage
            static_assert((LONG ? 8 : 4) * PS >= 4 * WT, "not enough store slots for a tile");
            PWS_STAGE(0, 0, 0, WS);
            PWS_STAGE(1, 1, PS, WS);
            PWS_STAGE(2, 2, 2 * PS, WS);
            PWS_STAGE(3, 3, 3 * PS, WS);
            if constexpr (LONG) {
                PWS_STAGE(4, 0, 4 * PS, WS);
                PWS_STAGE(5, 0, 5 * PS, WS);
                PWS_STAGE(6, 0, 6 * PS, WS);
                PWS_STAGE(7, 0, 7 * PS, WS);
            }
            for (int s = LONG ? 8 : 4; s < nst; s += 2) {
                stage_body(IC<SET>{}, IC<0>{}, IC<0>{}, IC<-1>{}, IC<1>{}, s, xc, true, xn, nlive, rn, yp);
                __syncthreads();
                stage_body(IC<SET>{}, IC<1>{}, IC<0>{}, IC<-1>{}, IC<1>{}, s + 1, xc, true, xn, nlive, rn, yp);
                __syncthreads();
            }
        } else if constexpr (LONG) {
            PWS_STAGE(0, 0, 0, 8);  // (after this barrier every wave has read its part of the residual copy: the
            PWS_STAGE(1, 1, 6, 8);  //  next tile's may be written)
            PWS_STAGE(2, 2, 12, 8);
            PWS_STAGE(3, 3, 18, 8);
            PWS_STAGE(4, 0, 24, 8);
            PWS_STAGE(5, 0, 30, 8);
            PWS_STAGE(6, 0, 36, 8);
            PWS_STAGE(7, 0, 42, 8);
            for (int s = 8; s < nst; s += 2) {
                stage_body(IC<SET>{}, IC<0>{}, IC<0>{}, IC<-1>{}, IC<1>{}, s, xc, true, xn, nlive, rn, yp);
                __syncthreads();
                stage_body(IC<SET>{}, IC<1>{}, IC<0>{}, IC<-1>{}, IC<1>{}, s + 1, xc, true, xn, nlive, rn, yp);
                __syncthreads();
            }
        } else {
            PWS_STAGE(0, 0, 0, 4);
            PWS_STAGE(1, 1, 12, 4);
            PWS_STAGE(2, 2, 24, 4);
            PWS_STAGE(3, 3, 36, 4);
            for (int s = 4; s < nst; s += 2) {
                stage_body(IC<SET>{}, IC<0>{}, IC<0>{}, IC<-1>{}, IC<1>{}, s, xc, true, xn, nlive, rn, yp);
                __syncthreads();
                stage_body(IC<SET>{}, IC<1>{}, IC<0>{}, IC<-1>{}, IC<1>{}, s + 1, xc, true, xn, nlive, rn, yp);
                __syncthreads();
            }
        }
#undef PWS_STAGE
        PWS_STAMP(i == 2);  // 1: K loop done
    };
    int i = 0;
    for (; i < n_my; i += 2) {
        tile_body(IC<0>{}, i);
        if (i + 1 >= n_my) break;
        tile_body(IC<1>{}, i + 1);
    }
    // ---- the last tile's stores (the 16-pass MFMA needs 18 wait states before its result is read; explicit as in
    // conv_mfma.hip): set 0 when the workgroup had an odd number of tiles
    const int last = n_my - 1;
    int r0s = r0_of(last);
    asm volatile("s_nop 15\n\ts_nop 7"
                 : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]),
                   "+s"(r0s));
    const RowRef yl = row_ref(p.y, p.N, r0s, wave_live);
    auto finish = [&](auto set_c) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value;
        if constexpr (WIDE) {
#pragma unroll
            for (int idx = 0; idx < 4 * WT; ++idx) store_quad(IC<SET>{}, idx, yl);
        } else {
#pragma unroll
            for (int q = 0; q < WT; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                {
                    __builtin_amdgcn_raw_buffer_store_b32(out_bits(IC<SET>{}, q, e),
                                                          rsrc_at(yl, 32 * q + (e & 3) + 8 * (e >> 2), p.N), evoff, 0, PWS_STORE_AUX);
                    if constexpr (EMIT) if (e == 15) emit_bits(IC<SET>{}, q, r0s, r0s < p.M);
                }
        }
    };
    if (last & 1) finish(IC<1>{});  // block-uniform
    else finish(IC<0>{});
    if (p.fw) {  // block-uniform
        const long long total = (long long)p.fN * p.fC;
        for (long long i = (long long)blockIdx.x * 256 + tid; i < total; i += (long long)gridDim.x * 256)
            if (p.fform) mpsr::wino3z_filter_one(p.fw, p.fN, p.fC, p.fu, i);
            else mpsr::wino3_filter_one(p.fw, p.fN, p.fC, p.fu, i);
    }
#ifdef PWS_TRACE
    if (p.trace && lane == 0) {
        unsigned long long *dst = p.trace + ((size_t)blockIdx.x * 4 + wave) * 16;
        for (int k = 0; k < 16; ++k) dst[k] = ts[k];
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// Fully-connected layers with few rows (the heads: M = batch = 256 rows, K ~ 1024, N = 1024 / 27 / 2; reference
// monopsr_output_builder.py:126-302).  A 64 x 64-tile GEMM makes 64 workgroups of them -- a quarter of the CUs, each
// running a 1000-deep K loop alone (36 us per layer).  Here a workgroup owns ONE 32 x 32 output tile (8 x 32 = 256
// workgroups for 256 x 1024) and its four waves split K; both operands' fragments come straight from global memory
// (lane = (row, k half), 16 bytes: the four 8-k chunks of a 128-byte line are requested together, eight chunks in
// flight per wave), no LDS in the K loop; the four partial tiles are summed through LDS and wave 0 applies bias /
// residual / ReLU.
// SPLIT = 1 (img_fc at the reference's 32 boxes per image: 32 rows x K = 18432 -> N = 2048): K is longer than four
// waves can stream (64 workgroups would read the 151 MB of weights), so blockIdx.y selects one of several K slabs and
// the workgroup stores its partial tile into slab blockIdx.y of a scratch tensor; winograd_finish_slices (winograd3z.hip)
// adds the slabs in order + bias + ReLU.  Every weight byte is read once per 32-row block: memory-bound.
template <int SPLIT>
__global__ __launch_bounds__(256) void fc_rows_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                      const float *__restrict__ bias, const float *__restrict__ residual,
                                                      float *__restrict__ y, int M, int N, int K, int relu,
                                                      unsigned xbytes, unsigned wbytes, int kslab)
{
    __shared__ __attribute__((aligned(16))) float part[3][16][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntn = (N + 31) >> 5;
    const int r0 = (blockIdx.x / ntn) * 32, n0 = (blockIdx.x % ntn) * 32;
    const int kper = kslab >> 2;        // the slab (= K unless SPLIT) a multiple of 32: every wave gets whole 8-k chunks
    const int nch = kper >> 3;          // chunks per wave
    const int i = lane & 31, h = lane >> 5;
    const unsigned OOB = 0x80000000u;
    const unsigned koff = (unsigned)((SPLIT ? (int)blockIdx.y * kslab : 0) + wave * kper + 4 * h) * 4u;
    const unsigned aoff = r0 + i < M ? (unsigned)(r0 + i) * (unsigned)K * 4u + koff : OOB;
    const unsigned boff = n0 + i < N ? (unsigned)(n0 + i) * (unsigned)K * 4u + koff : OOB;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w), 0, (int)wbytes, 0x00020000);
    constexpr int PF = 8;  // chunks in flight
    float4 fa[PF], fb[PF];
    auto request = [&](int c, int slot) __attribute__((always_inline)) {
        const bool live = c < nch;
        const __amdgpu_buffer_rsrc_t rxl =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, live ? (int)xbytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rwl =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w), 0, live ? (int)wbytes : 0, 0x00020000);
        fa[slot] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rxl, aoff, (live ? c : 0) * 32, 0));
        fb[slot] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rwl, boff, (live ? c : 0) * 32, 0));
    };
    (void)rx;
    (void)rw;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int c = 0; c < PF; ++c) request(c, c);
    for (int c0 = 0; c0 < nch; c0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const float4 a = fa[u], b = fb[u];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            request(c0 + PF + u, u);
        }
    }
    // (the 16-pass MFMA needs 18 wait states before its result is read; explicit as in conv_mfma.hip)
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) part[wave - 1][e][lane] = acc[e];
    }
    __syncthreads();
    if (wave > 0) return;
    // accumulator element e of a lane: row (e & 3) + 8 (e >> 2) + 4 (lane >> 5), column lane & 31
    const int col = n0 + i;
    const float bv = (!SPLIT && bias && col < N) ? bias[col] : 0.f;
    float *yo = SPLIT ? y + (size_t)blockIdx.y * M * N : y;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = r0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        float v = ((acc[e] + part[0][e][lane]) + part[1][e][lane]) + part[2][e][lane] + bv;
        if (row < M && col < N) {
            if (!SPLIT && residual) v += residual[(size_t)row * N + col];
            if (!SPLIT && relu) v = fmaxf(v, 0.f);
            yo[(size_t)row * N + col] = v;
        }
    }
}

std::atomic<int> g_pw_override{-1};  // -1 heuristic, 0 never, 1 wherever it applies
std::atomic<int> g_pws_per_cu{2};
unsigned long long *g_pw_trace = nullptr;

}  // namespace

extern "C" void mpsr_debug_set_conv_pointwise(int mode) { g_pw_override = mode; }
extern "C" void mpsr_debug_set_pointwise_per_cu(int n) { g_pws_per_cu = n > 0 ? n : 2; }
extern "C" void mpsr_debug_set_pointwise_trace(void *buf) { g_pw_trace = static_cast<unsigned long long *>(buf); }

namespace mpsr {

int pointwise_override() { return g_pw_override.load(); }

// Shapes the kernel takes: K a multiple of 64 and >= 128 (whole pairs of 32-k stages, four of them carry the residual
// and store duties), N a multiple of 32, 32-bit byte offsets, few enough column blocks for a persistent grid whose
// stride keeps a workgroup on one column block.
bool pointwise_applies(long long M, int K, int N)
{
    return M > 0 && K >= 128 && K % 64 == 0 && N >= 32 && N % 32 == 0 && (N + 127) / 128 <= 32 &&
           M * K * 4 < 0x7f000000LL && M * N * 4 < 0xfffffff0LL && (long long)N * K * 4 < 0x7f000000LL;
}

// Shapes whose result can leave through a ReLU bit mask (conv1x1_pointwise_masked): the long-K instantiations
bool pointwise_masked_applies(long long M, int K, int N)
{
    return !pws::WIDE && pointwise_applies(M, K, N) && K / pws::KS >= 8;
}

static int pointwise_launch(const float *x, long long M, int K, const float *w, const float *bias, const float *residual,
                            int relu, const unsigned *mask, unsigned *emit, float *y, int N, hipStream_t s);

int conv1x1_pointwise(const float *x, long long M, int K, const float *w, const float *bias,
                             const float *residual, int relu, float *y, int N, hipStream_t s)
{
    MPSR_REQUIRE(pointwise_applies(M, K, N), "conv1x1_pointwise: unsupported shape (M=%lld K=%d N=%d)", M, K, N);
    return pointwise_launch(x, M, K, w, bias, residual, relu, nullptr, nullptr, y, N, s);
}

// conv1x1_pointwise that also writes the ReLU bit mask of its result (bits of rows >= M unspecified)
int conv1x1_pointwise_emit(const float *x, long long M, int K, const float *w, const float *bias, const float *residual,
                           int relu, float *y, unsigned *bits, int N, hipStream_t s)
{
    MPSR_REQUIRE(pointwise_masked_applies(M, K, N), "conv1x1_pointwise_emit: unsupported shape (M=%lld K=%d N=%d)", M, K, N);
    MPSR_REQUIRE(bits, "conv1x1_pointwise_emit: null mask");
    return pointwise_launch(x, M, K, w, bias, residual, relu, nullptr, bits, y, N, s);
}

// y[m][n] = keep(m, n) ? sum_k x[m][k] w[n][k] + bias[n] + residual[m][n] : 0, keep = bit (m & 31) of
// mask[(m >> 5) * N + n] ((M + 31) / 32 * N words, mpsr_relu_bitmask)
int conv1x1_pointwise_masked(const float *x, long long M, int K, const float *w, const float *bias, const float *residual,
                             const unsigned *mask, float *y, int N, hipStream_t s)
{
    MPSR_REQUIRE(pointwise_masked_applies(M, K, N), "conv1x1_pointwise_masked: unsupported shape (M=%lld K=%d N=%d)", M, K, N);
    MPSR_REQUIRE(mask, "conv1x1_pointwise_masked: null mask");
    return pointwise_launch(x, M, K, w, bias, residual, 0, mask, nullptr, y, N, s);
}

static int pointwise_launch(const float *x, long long M, int K, const float *w, const float *bias, const float *residual,
                            int relu, const unsigned *mask, unsigned *emit, float *y, int N, hipStream_t s)
{
    using namespace pws;
    PwsParams p;
    p.mask = mask;
    p.emit = emit;
    p.maskbytes = (unsigned)(((M + 31) / 32) * N * 4);
    p.x = x; p.w = w; p.bias = bias; p.residual = residual; p.y = y;
    p.M = (int)M; p.N = N; p.K = K; p.relu = relu;
    p.rgroups = (int)((M + ROWS - 1) / ROWS);
    p.cblocks = ceil_div(N, COLS);
    p.ntiles = ceil_div(p.rgroups, 8) * 8 * p.cblocks;
    p.wbytes = (unsigned)((long long)N * K * 4);
    // CU count of the CURRENT device, cached per device id (a process may drive several GPUs; the count only sizes
    // the persistent grid)
    static std::atomic<int> cu_count[64];
    int dev = 0;
    MPSR_CHECK_HIP(hipGetDevice(&dev));
    int cus = dev >= 0 && dev < 64 ? cu_count[dev].load() : 0;
    if (cus == 0) {
        int n = 256;
        MPSR_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        cus = n > 0 ? n : 256;
        if (dev >= 0 && dev < 64) cu_count[dev] = cus;
    }
    // persistent grid: g_pws_per_cu workgroups per CU, rounded down to a multiple of 8 x column blocks
    const int unit = 8 * p.cblocks;
    int grid = g_pws_per_cu.load() * cus / unit * unit;
    if (grid < unit) grid = unit;
    if (grid > p.ntiles) grid = p.ntiles;
    p.trace = g_pw_trace;
    // a pending filter-transform job rides on this launch
    p.fw = g_filter_tail_job.w; p.fu = g_filter_tail_job.u; p.fN = g_filter_tail_job.N; p.fC = g_filter_tail_job.C;
    p.fform = g_filter_tail_job.form;
    if (p.fw) g_filter_tail_done = g_filter_tail_job;
    g_filter_tail_job = FilterTailJob();
    const size_t lds_bytes = (size_t)LDS_B;
#define MPSR_PW(...)                                                                                                  \
    do {                                                                                                              \
        MPSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pw_conv_kernel<__VA_ARGS__>),               \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));              \
        hipLaunchKernelGGL((pw_conv_kernel<__VA_ARGS__>), dim3(grid), dim3(256), lds_bytes, s, p);                    \
    } while (0)
#if !PWS_WIDE
    if (mask || emit) {
        if (mask && residual) MPSR_PW(true, true, true);
        else if (mask) MPSR_PW(false, true, true);
        else if (residual) MPSR_PW(true, true, false, true);
        else MPSR_PW(false, true, false, true);
        MPSR_CHECK_LAUNCH("pw_conv_kernel");
        return MPSR_OK;
    }
#endif
    if (K / KS >= 8) {
        if (residual) MPSR_PW(true, true);
        else MPSR_PW(false, true);
    } else {
        if (residual) MPSR_PW(true, false);
        else MPSR_PW(false, false);
    }
#undef MPSR_PW
    MPSR_CHECK_LAUNCH("pw_conv_kernel");
    return MPSR_OK;
}


// Few-row fully-connected layers (fc_rows_kernel): K a multiple of 32 up to 4096 (longer K: the stream-K kernel, whose
// tiles share the operand traffic), 32-bit byte offsets.
bool fc_rows_applies(long long M, int K, int N)
{
    return M > 0 && M <= 2048 && K >= 128 && K <= 4096 && K % 32 == 0 && N > 0 && M * K * 4 < 0x7f000000LL &&
           (long long)N * K * 4 < 0x7f000000LL;
}

int fc_rows(const float *x, long long M, int K, const float *w, const float *bias, const float *residual, int relu,
            float *y, int N, hipStream_t s)
{
    MPSR_REQUIRE(fc_rows_applies(M, K, N), "fc_rows: unsupported shape (M=%lld K=%d N=%d)", M, K, N);
    const unsigned grid = (unsigned)(((M + 31) / 32) * ((N + 31) / 32));
    hipLaunchKernelGGL(fc_rows_kernel<0>, dim3(grid), dim3(256), 0, s, x, w, bias, residual, y, (int)M, N, K, relu,
                       (unsigned)(M * K * 4), (unsigned)((long long)N * K * 4), K);
    MPSR_CHECK_LAUNCH("fc_rows_kernel");
    return MPSR_OK;
}

// winograd3z.hip
int winograd_finish_slices(const float *part, const float *bias, float *y, size_t y_floats, int nslices, int N, int relu,
                           hipStream_t s);

std::atomic<int> g_fc_split_rows{96};  // mpsr_debug_set_fc_split_rows: the K-split few-row kernel up to this many rows (0: never)

// K slabs of the split form for a long-K layer with few rows: enough workgroups to stream the weights from every CU
static int fc_rows_slabs(long long M, int K, int N)
{
    const long long tiles = ((M + 31) / 32) * ((N + 31) / 32);
    int s = 1;
    while (tiles * s < 512 && s < 16 && K % (64 * s) == 0) s *= 2;
    return s;
}

// img_fc at small batches: few rows, K too long for the four waves of one workgroup (fc_rows_applies stops at 4096)
bool fc_rows_split_applies(long long M, int K, int N, const float *bias, const float *y, const float *ws, size_t ws_floats)
{
    const int rows = g_fc_split_rows.load();
    if (!(M > 0 && M <= rows && K > 4096 && K % 64 == 0 && N > 0 && N % 4 == 0 && ws && M * K * 4 < 0x7f000000LL &&
          (long long)N * K * 4 < 0x7f000000LL))
        return false;
    const int slabs = fc_rows_slabs(M, K, N);
    return slabs > 1 && (size_t)slabs * M * N <= ws_floats && ((uintptr_t)ws & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
           (!bias || ((uintptr_t)bias & 15) == 0);
}

int fc_rows_split(const float *x, long long M, int K, const float *w, const float *bias, int relu, float *y, int N,
                  float *ws, hipStream_t s)
{
    const int slabs = fc_rows_slabs(M, K, N);
    const dim3 grid((unsigned)(((M + 31) / 32) * ((N + 31) / 32)), (unsigned)slabs);
    hipLaunchKernelGGL(fc_rows_kernel<1>, grid, dim3(256), 0, s, x, w, nullptr, nullptr, ws, (int)M, N, K, 0,
                       (unsigned)(M * K * 4), (unsigned)((long long)N * K * 4), K / slabs);
    MPSR_CHECK_LAUNCH("fc_rows_kernel");
    return winograd_finish_slices(ws, bias, y, (size_t)M * N, slabs, N, relu, s);
}

}  // namespace mpsr

extern "C" void mpsr_debug_set_fc_split_rows(int rows) { mpsr::g_fc_split_rows = rows; }
