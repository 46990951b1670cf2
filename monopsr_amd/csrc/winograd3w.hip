// Winograd F(3x3,3x3) for the atrous 3x3 layers whose pixel sub-grids are single 3x3 tiles (ResNet-101 block3's conv2 at
// output stride 4: 12x12 maps, dilation 4, 23 launches per step; reference graph object_detection/nets/resnet_v1.py:116-127,
// resnet_utils.py:194-196), in the form where ONE WAVE OWNS ALL 25 POSITIONS of its (32 tiles x 32 output channels) block.
//
// winograd3.hip spreads a tile's 25 positions over eight waves (6-7 positions each), so the output transform A^T M A
// needs every wave's accumulators: a 100 KB exchange through LDS in two rounds, and a workgroup lives for only 32 short
// K steps around it (measured: ~25 % of the launch is prologue / exchange / epilogue).  Here a wave keeps 25 x 16 = 400
// accumulator registers -- 16 positions in the accumulator half of the register file, 9 in the vector half, one wave per
// SIMD -- every position's 32x32 tile has the same lane layout, so the 25 values of one (tile, channel) sit in ONE LANE
// and the output transform is lane-local: no exchange, no second round, and each wave issues 4x the MFMAs per epilogue.
//
// Workgroup = 4 waves = 32 tiles x 128 output channels (WM = 1; WM = 2: 64 tiles x 64 channels).  The transformed patches
// (A operand) of a K step (8 channels) are produced once per workgroup -- one patch per thread and step at WM = 1 -- and
// shared through LDS by the four waves, i.e. the 56 vector instructions + 25 LDS stores of a patch are amortised over 128
// output channels (every vector instruction next to an fp32 MFMA costs ~4 cycles of the SIMD's matrix time, DESIGN 4.1
// finding 6: the transform is THE overhead of this kernel, 1 patch per 100 MFMAs here against 1 per 52 in winograd3.hip).
// B fragments come straight from the transformed filters (L2) as in winograd3.hip, same layout (wino3_filter.h), so the
// filter cache and the tail-job transform serve both kernels, and the accumulation order per output is the same: the two
// kernels return identical bits.
//
// K step = 100 MFMAs (position-major, 4 per position), double-buffered A with ONE barrier per step at slot 92: a step's
// last A read (position 24) is issued at slot 88, the next step's first at slot 92 right behind the barrier, the producer's
// stores of the next step's patches sit in slots 0..27 (they target the buffer whose reads ended at the previous barrier).
// A fragments two positions ahead in a ring of four register quads (25 is not a multiple of 3: the ring colours are
// 0 1 2 x 7, then 0 1 2 3), B fragments seven positions (28 MFMAs, ~1800 cycles) ahead in a ring of nine (colours
// 0..7, 0..7, 0..8): inside a step the transformed filters of a layer come from the Infinity Cache, not from L2, and a
// wave alone on its SIMD has nobody to hide a late fragment behind.
#include <type_traits>
#include <utility>

#include "common.h"
#include "wino3_filter.h"
#include "wino3_transforms.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using mpsr::FastDiv;
using mpsr::fdiv;
using mpsr::w3t::at3;
using mpsr::w3t::bt5;

namespace f3w {
constexpr int KC = 8, NP = 25;
constexpr unsigned OOB = 0x80000000u;
constexpr int cA(int p) { return p < 21 ? p % 3 : p - 21; }  // ring colour of position p's A fragment (4 quads)
#ifndef W3W_BPRE
#define W3W_BPRE 7
#endif
constexpr int BPRE = W3W_BPRE;  // B fragments requested this many positions ahead: 4 (ring of 5), 7 (ring of 9), 11 (ring of 13)
constexpr int BRING = BPRE == 4 ? 5 : BPRE == 7 ? 9 : 13;
static_assert(BPRE == 4 || BPRE == 7 || BPRE == 11, "ring colourings: 25 = 5 x 5, 8 + 8 + 9, 12 + 13");
constexpr int cB(int p)  // ... of its B fragment
{
    return BPRE == 4 ? p % 5 : BPRE == 7 ? (p < 16 ? p % 8 : p - 16) : (p < 12 ? p : p - 12);
}
}  // namespace f3w

struct Wino3WParams {
    const float *x, *u, *bias, *mask;
    float *y;
    int H, W, C, N, dil, T;  // T = B * dil * dil tiles (one per pixel sub-grid)
    int cblocks, nblocks, mblocks, relu;
    unsigned xbytes, ubytes, ybytes;
    FastDiv div_tpi, div_d;  // tiles per image = dil^2, dil
};

#ifdef W3W_TRACE  // (timing builds only: tools/wino3w_trace.py) cycle stamps of the first eight workgroups' waves
__device__ unsigned long long g_w3w_trace[8 * 4 * 40];
#define W3W_STAMP(i)                                                                                        \
    do {                                                                                                    \
        if (blockIdx.x < 8 && (threadIdx.x & 63) == 0)                                                      \
            g_w3w_trace[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 40 + (i)] = __builtin_readcyclecounter();  \
    } while (0)
#else
#define W3W_STAMP(i) do { } while (0)
#endif

template <int V>
using ICW = std::integral_constant<int, V>;
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f)
{
    (f(ICW<Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// Positions 0..15 accumulate in the accumulator half of the register file under LITERAL names (position q = a[16 q : 16 q + 15]):
// hipcc allocates MFMA results of a 512-register kernel to that half only and copies whole 16-register tuples to read one
// element, so compiler-managed tuples there either spill (400 > 256) or turn the lane-local epilogue into ~1000 register
// moves.  The names are the kernel's own by the clobber list of W3W_CLAIM_ACC (which also makes the kernel descriptor
// allocate all 256); hipcc touches that half only to spill vector registers, which this kernel must never do:
// tests/test_build_audit.py checks the compiled kernel for v_accvgpr_* instructions outside these statements.
// (no wait states inside the statements: a vector-ALU write of an A / B operand needs two before the MFMA that reads it,
// which hipcc does not pad for inline asm -- the operands here are written by LDS / buffer loads only, and
// tests/test_build_audit.py checks the compiled kernel for a vector-ALU write of an operand in the two instructions in
// front of each MFMA; a blanket s_nop 1 measured 1 % of the K loop)
#ifdef W3W_PAD_NOP
#define W3W_PAD "s_nop 1\n\t"
#else
#define W3W_PAD ""
#endif
#define W3W_MFMA_A(q, a, b)                                                                                   \
    asm volatile(W3W_PAD "v_mfma_f32_32x32x2_f32 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(16 * (q)), \
                 "i"(16 * (q) + 15))
#define W3W_MFMA_V(acc, a, b) asm volatile(W3W_PAD "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define W3W_ZERO16(b)                                                                                                     \
    asm volatile("v_accvgpr_write_b32 a%c0, 0\n\tv_accvgpr_write_b32 a%c1, 0\n\tv_accvgpr_write_b32 a%c2, 0\n\t"          \
                 "v_accvgpr_write_b32 a%c3, 0\n\tv_accvgpr_write_b32 a%c4, 0\n\tv_accvgpr_write_b32 a%c5, 0\n\t"          \
                 "v_accvgpr_write_b32 a%c6, 0\n\tv_accvgpr_write_b32 a%c7, 0\n\tv_accvgpr_write_b32 a%c8, 0\n\t"          \
                 "v_accvgpr_write_b32 a%c9, 0\n\tv_accvgpr_write_b32 a%c10, 0\n\tv_accvgpr_write_b32 a%c11, 0\n\t"        \
                 "v_accvgpr_write_b32 a%c12, 0\n\tv_accvgpr_write_b32 a%c13, 0\n\tv_accvgpr_write_b32 a%c14, 0\n\t"       \
                 "v_accvgpr_write_b32 a%c15, 0" ::"i"((b)), "i"((b) + 1), "i"((b) + 2), "i"((b) + 3), "i"((b) + 4),       \
                 "i"((b) + 5), "i"((b) + 6), "i"((b) + 7), "i"((b) + 8), "i"((b) + 9), "i"((b) + 10), "i"((b) + 11),      \
                 "i"((b) + 12), "i"((b) + 13), "i"((b) + 14), "i"((b) + 15))
#define W3W_READ_ACC(dst, idx) asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(dst) : "i"(idx))
#define W3W_CLAIM_ACC()                                                                                                   \
    asm volatile("" :: : \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
    "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
    "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
    "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
    "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
    "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
    "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
    "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", \
    "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", \
    "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", \
    "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", \
    "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", \
    "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", \
    "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", \
    "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", \
    "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255")

// MASK: a data-gradient launch of the training path (p.mask = a tensor shaped like y; an output is kept where the mask is
// positive: the ReLU gradient of the layer the gradient belongs to, applied in the store path)
template <int WM, bool MASK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino3w_conv_kernel(const Wino3WParams p)
{
    using namespace f3w;
    constexpr int WN = 4 / WM, MT = 32 * WM, NT = 32 * WN;
    constexpr int APOS = MT * KC;    // floats per position of an A buffer
    constexpr int ABUF = NP * APOS;  // one A buffer (WM = 1: 25.6 KB)
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int xcd = blockIdx.x & 7, l_ = blockIdx.x >> 3;
    const int nb = l_ % p.nblocks;
    const int mb = (l_ / p.nblocks) * 8 + xcd;
    if (mb >= p.mblocks) return;  // block-uniform
    const int n0 = nb * NT, t0 = mb * MT;
    W3W_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wave % WM, ni = wave / WM;
    const int nsteps = p.cblocks, d = p.dil, tpi = d * d;

    // ---- A producer: thread = (tile, channel of the step) for WM patches per step; nine 4-byte requests per patch
    const int ch = lane & 7;
    unsigned abase[WM];
    float *awr[WM];
#pragma unroll
    for (int r = 0; r < WM; ++r) {
        const int lt = 32 * r + 8 * wave + (lane >> 3);
        const int t = t0 + lt;
        const int img = fdiv(t, p.div_tpi), sub = t - img * tpi;
        const int a = fdiv(sub, p.div_d), b = sub - a * d;
        abase[r] = t < p.T ? (unsigned)(((img * p.H + a) * p.W + b) * p.C + ch) * 4u : OOB;
        // A[buf][pos][tile][8 channels], 16-byte halves swapped on odd 8-row blocks
        awr[r] = lds + lt * 8 + 4 * ((ch >> 2) ^ ((lt >> 3) & 1)) + (ch & 3);
    }
    float raw[WM][9];  // column by column: raw[3 j + i] = sub-grid pixel (row i, column j)
    // (requests past the last K step are not special-cased: they read the neighbouring channels / positions or fall outside
    // the descriptor's range and return zeros; nothing consumes them)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.xbytes, 0x00020000);
    auto load_raw = [&](int step, auto rc, auto Lc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value, L = decltype(Lc)::value, j = L / 3, i = L % 3;
        const unsigned so = (unsigned)((d * i * p.W + d * j) * p.C + step * KC) * 4u;
        raw[r][L] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, abase[r], so, 0));
    };
    float pa[25];  // the 5x5 transformed patch being built (one patch at a time)
    auto vertical = [&](auto rc, auto jc) __attribute__((always_inline)) {  // data column j -> the five rows of patch column j + 1
        constexpr int r = decltype(rc)::value, j = decltype(jc)::value;
        bt5(raw[r][3 * j], raw[r][3 * j + 1], raw[r][3 * j + 2], pa[j + 1], pa[5 + j + 1], pa[10 + j + 1], pa[15 + j + 1],
            pa[20 + j + 1]);
    };
    auto horizontal = [&](auto ic) __attribute__((always_inline)) {  // row i: three values -> five
        constexpr int i = decltype(ic)::value;
        float t0_, t1_, t2_, t3_, t4_;
        bt5(pa[5 * i + 1], pa[5 * i + 2], pa[5 * i + 3], t0_, t1_, t2_, t3_, t4_);
        pa[5 * i] = t0_;
        pa[5 * i + 1] = t1_;
        pa[5 * i + 2] = t2_;
        pa[5 * i + 3] = t3_;
        pa[5 * i + 4] = t4_;
    };
    auto store_a = [&](auto rc, int buf, int pos) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        awr[r][buf * ABUF + pos * APOS] = pa[pos];
    };
    // producer duty of slot m of a K step: patch r transformed in slots 14 r .. 14 r + 13 (3 columns, then a row every other
    // slot with its five stores behind it), the next-but-one step's nine requests per patch from slot 30 on
    auto duty = [&](int s, auto bufc, auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value, nbuf = decltype(bufc)::value ^ 1;
#ifdef W3W_SKIP_PROD  // (timing experiments only: wrong results)
        return;
#endif
        static_for<WM>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value, o = m - 14 * r;
            if constexpr (o >= 0 && o < 3) vertical(rc, ICW<o>{});
            if constexpr (o >= 3 && o < 13 && ((o - 3) & 1) == 0) horizontal(ICW<(o - 3) / 2>{});
            if constexpr (o >= 4 && o < 14) {
                constexpr int i = (o - 4) / 2;
                if constexpr (((o - 4) & 1) == 0) {
                    store_a(rc, nbuf, 5 * i);
                    store_a(rc, nbuf, 5 * i + 1);
                    store_a(rc, nbuf, 5 * i + 2);
                } else {
                    store_a(rc, nbuf, 5 * i + 3);
                    store_a(rc, nbuf, 5 * i + 4);
                }
            }
            constexpr int L = m - 30 - 9 * r;
            if constexpr (L >= 0 && L < 9) load_raw(s + 2, rc, ICW<L>{});
        });
    };

    // ---- B fragments from the transformed filters U[cb][pos][n][8], lane = (n = lane & 31, k half = lane >> 5)
    const int nB = n0 + 32 * ni + (lane & 31);
    const unsigned bvoff = nB < p.N ? (unsigned)(nB * KC + 4 * (lane >> 5)) * 4u : OOB;
    const unsigned bpstride = (unsigned)p.N * KC * 4u;
    float4 fb[BRING];
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.u), 0, (int)p.ubytes, 0x00020000);
    auto load_b = [&](int step, auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
#ifdef W3W_SKIP_BLOAD
        if (step > 0) return;
#endif
        const unsigned so = ((unsigned)step * (unsigned)NP + (unsigned)q) * bpstride;
        fb[cB(q)] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ru, bvoff, so, 0));
    };
    // ---- A fragments: lane = (tile = lane & 31 of the wave's 32, k half = lane >> 5)
    const float *ard = lds + (32 * mi + (lane & 31)) * 8 + 4 * ((lane >> 5) ^ (((lane & 31) >> 3) & 1));
    float4 fa[4];
    auto load_a = [&](int buf, auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
#ifdef W3W_SKIP_ALOAD
        if (buf >= 0 && q > 1) return;
#endif
        fa[cA(q)] = *reinterpret_cast<const float4 *>(ard + buf * ABUF + q * APOS);
    };

    f32x16 accV[9];  // positions 16..24, in the vector half of the register file (0..15: a[0:255] by name)
    auto mfma = [&](auto qc, float av, float bv) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        if constexpr (q < 16) {
            W3W_MFMA_A(q, av, bv);
        } else {
            f32x16 &ac = accV[q - 16];  // (a reference first: an asm operand alone does not make the lambda capture the array)
            W3W_MFMA_V(ac, av, bv);
        }
    };

    // ---- prologue: requests of step 0, the first B fragments, accumulators, A of step 0, requests of step 1
    static_for<WM>([&](auto rc) __attribute__((always_inline)) {
        static_for<9>([&](auto Lc) __attribute__((always_inline)) { load_raw(0, rc, Lc); });
    });
    static_for<BPRE>([&](auto qc) __attribute__((always_inline)) { load_b(0, qc); });
    W3W_CLAIM_ACC();
    static_for<16>([&](auto qc) __attribute__((always_inline)) { W3W_ZERO16(16 * decltype(qc)::value); });
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) accV[q][e] = 0.f;
    static_for<WM>([&](auto rc) __attribute__((always_inline)) {
        static_for<3>([&](auto jc) __attribute__((always_inline)) { vertical(rc, jc); });
        static_for<5>([&](auto ic) __attribute__((always_inline)) { horizontal(ic); });
#pragma unroll
        for (int i = 0; i < 25; ++i) store_a(rc, 0, i);
        static_for<9>([&](auto Lc) __attribute__((always_inline)) { load_raw(1, rc, Lc); });
    });
    __syncthreads();
    load_a(0, ICW<0>{});
    load_a(0, ICW<1>{});
    W3W_STAMP(1);

    // ---- K loop: slot m = MFMA k = m % 4 of position q = m / 4, followed by the slot's loads and producer work
    auto kstep = [&](int s, auto bufc) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        static_for<100>([&](auto mc) __attribute__((always_inline)) {
            constexpr int m = decltype(mc)::value, q = m / 4, k = m % 4;
            const float av = k == 0 ? fa[cA(q)].x : k == 1 ? fa[cA(q)].y : k == 2 ? fa[cA(q)].z : fa[cA(q)].w;
            const float bv = k == 0 ? fb[cB(q)].x : k == 1 ? fb[cB(q)].y : k == 2 ? fb[cB(q)].z : fb[cB(q)].w;
            mfma(ICW<q>{}, av, bv);
            if constexpr (k == 0) {
#ifndef W3W_NO_BARRIER
                if constexpr (m == 92) __syncthreads();
#endif
                if constexpr (q + 2 < NP) load_a(buf, ICW<(q + 2) % NP>{});
                else load_a(buf ^ 1, ICW<(q + 2) % NP>{});
            }
            if constexpr (k == 1) {
                if constexpr (q + BPRE < NP) load_b(s, ICW<(q + BPRE) % NP>{});
                else load_b(s + 1, ICW<(q + BPRE) % NP>{});
            }
            duty(s, bufc, mc);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    for (int s = 0; s < nsteps; s += 2) {
        kstep(s, ICW<0>{});
        if (s < 32) W3W_STAMP(2 + s);
        kstep(s + 1, ICW<1>{});
        if (s < 32) W3W_STAMP(3 + s);
    }
    // the last MFMAs' results: 18 wait states before anything reads them (hipcc pads nothing behind inline asm)
    W3W_STAMP(34);
    int lane2 = lane;
    asm volatile("s_nop 15\n\ts_nop 7"
                 : "+v"(accV[0]), "+v"(accV[1]), "+v"(accV[2]), "+v"(accV[3]), "+v"(accV[4]), "+v"(accV[5]), "+v"(accV[6]),
                   "+v"(accV[7]), "+v"(accV[8]), "+v"(lane2));

    // ---- epilogue, lane-local: register e of every position belongs to tile (e & 3) + 8 (e >> 2) + 4 (lane >> 5) of the
    // wave's 32 and to channel lane & 31: A^T M A on the lane's own 25 values, bias, ReLU / mask, nine strided pixels out
    const int n = n0 + 32 * ni + (lane2 & 31);
    const float bias_v = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    const float floor_v = p.relu ? 0.f : -__builtin_inff();  // ReLU as one max
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.mask), 0, MASK ? (int)p.ybytes : 0, 0x00020000);
    unsigned so[9];  // byte offset of output pixel (i, j) of a tile from its pixel (0, 0): so[3 j + i]
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) so[3 * j + i] = (unsigned)((d * i * p.W + d * j) * p.N) * 4u;
    static_for<16>([&](auto ec) __attribute__((always_inline)) {
        constexpr int e = decltype(ec)::value;
        const int t = t0 + 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * (lane2 >> 5);
        const int img = fdiv(t, p.div_tpi), sub = t - img * tpi;
        const int a = fdiv(sub, p.div_d), b = sub - a * d;
        const unsigned voff = (t < p.T && n < p.N) ? (unsigned)(((img * p.H + a) * p.W + b) * p.N + n) * 4u : OOB;
        float mk[9];
        if constexpr (MASK) {
#pragma unroll
            for (int o = 0; o < 9; ++o)
                mk[o] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, voff, so[o], 0));
        }
        float z[5][3];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            float mv[5];
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                const int q = 5 * u + v;
                if (q < 16) W3W_READ_ACC(mv[v], 16 * q + e);
                else mv[v] = accV[q >= 16 ? q - 16 : 0][e];
            }
            at3(mv[0], mv[1], mv[2], mv[3], mv[4], z[u][0], z[u][1], z[u][2]);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float yv[3];
            at3(z[0][j], z[1][j], z[2][j], z[3][j], z[4][j], yv[0], yv[1], yv[2]);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float v = fmaxf(yv[i] + bias_v, floor_v);
                if constexpr (MASK) v = mk[3 * j + i] > 0.f ? v : 0.f;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ry, voff, so[3 * j + i], 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // (one register column at a time: hipcc otherwise hoists all 400 reads)
    });
    W3W_STAMP(35);
}

}  // namespace

namespace mpsr {

// the one-wave-per-tile-block form applies where a sub-grid is one tile (th == 1); worth it once its workgroups fill the chip
bool winograd3w_applies(int B, int H, int W, int C, int N, int dilation)
{
    return dilation >= 1 && H == W && H == 3 * dilation && C % 16 == 0 && C >= 16 && N >= 1 &&
           (long long)B * H * W * N * 4 < 0x7ff00000LL;
}

// workgroups of a launch (32 tiles x 128 channels each)
long long winograd3w_workgroups(int B, int N, int dilation)
{
    return (long long)ceil_div(B * dilation * dilation, 32) * ceil_div(N, 128);
}

// launches the kernel on already transformed filters `u` (layout of wino3_filter.h).  Only WM = 1 (32 tiles x 128 channels
// per workgroup) is instantiated: at WM = 2 (64 x 64, two patches per thread) hipcc runs out of vector registers and
// parks values in a0..a8, i.e. inside position 0's accumulators.
int launch_winograd3w(const float *x, int B, int H, int W, int C, const float *u, const float *bias, int relu, float *y,
                      int N, int dilation, hipStream_t s, const float *mask)
{
    constexpr int wm = 1;
    using namespace f3w;
    Wino3WParams p;
    p.x = x; p.u = u; p.bias = bias; p.mask = mask; p.y = y;
    p.H = H; p.W = W; p.C = C; p.N = N; p.dil = dilation;
    p.T = B * dilation * dilation;
    p.cblocks = C / KC;
    const int MT = 32 * wm, NT = 32 * (4 / wm);
    p.nblocks = ceil_div(N, NT);
    p.mblocks = ceil_div(p.T, MT);
    p.relu = relu;
    p.xbytes = (unsigned)((long long)B * H * W * C * 4);
    p.ubytes = (unsigned)((size_t)NP * N * C * 4);
    p.ybytes = (unsigned)((long long)B * H * W * N * 4);
    p.div_tpi = make_fastdiv(dilation * dilation);
    p.div_d = make_fastdiv(dilation);
    const long long blocks = 8LL * ceil_div(p.mblocks, 8) * p.nblocks;
    if (blocks > 0x7fffffffLL) return fail(MPSR_ERR_UNSUPPORTED, "conv3x3_winograd3: grid too large");
    const size_t ldsb = (size_t)2 * NP * MT * KC * sizeof(float);
    const void *kern = mask ? reinterpret_cast<const void *>(wino3w_conv_kernel<1, true>)
                            : reinterpret_cast<const void *>(wino3w_conv_kernel<1, false>);
    MPSR_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    if (mask) hipLaunchKernelGGL((wino3w_conv_kernel<1, true>), dim3((unsigned)blocks), dim3(256), ldsb, s, p);
    else hipLaunchKernelGGL((wino3w_conv_kernel<1, false>), dim3((unsigned)blocks), dim3(256), ldsb, s, p);
    MPSR_CHECK_LAUNCH("wino3w_conv_kernel");
    return MPSR_OK;
}

}  // namespace mpsr

#ifdef W3W_TRACE
extern "C" int mpsr_debug_wino3w_trace(unsigned long long *host_out, int count)
{
    if (count > 8 * 4 * 40) count = 8 * 4 * 40;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_w3w_trace), sizeof(unsigned long long) * count) == hipSuccess ? 0 : 1;
}
#endif
