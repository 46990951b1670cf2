// The atrous 3x3 layers whose pixel sub-grids are single 3x3 tiles (ResNet-101 block3's conv2 at output stride 4: 12x12
// maps, dilation 4, 23 launches per step; reference graph object_detection/nets/resnet_v1.py:116-127,
// resnet_utils.py:194-196) in SIXTEEN products per (channel pair, tile).
//
// Such a sub-grid reads nothing outside itself (everything a tap reaches beyond it is SAME-padding zeros), so in one
// dimension its outputs are the MIDDLE three coefficients of the product of two quadratics -- a bilinear map of rank 4,
// not the 5 that F(3,3) spends on a general 5-point input (derivation, matrices and error: wino3_transforms.h).  Nested
// in two dimensions: Y (3x3) = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A with 4x4 = 16 element positions where
// F(3x3,3x3) (winograd3.hip, winograd3w.hip) has 25 and the direct form 81; transform constants 2, 3/2, 1/2: the fp32
// error of a direct convolution (5e-7 of the tensor scale, 1.8e-4 element-wise on heavy-tailed maps, where F(3x3,3x3)
// measures 4e-6 / 1.2e-3).  36 % fewer MFMAs, a cheaper input transform (42 operations + 16 LDS stores per patch instead of
// 56 + 25), and 16 x 16 = 256 accumulator registers: exactly the accumulator half of the register file.
//
// Kernel = winograd3w.hip's form: ONE WAVE OWNS ALL POSITIONS of its (32 tiles x 32 output channels) block, the output
// transform is lane-local; workgroup = 4 waves = 32 tiles x 128 channels sharing the transformed patches through LDS
// (double buffered), one patch per thread and K step of 8 channels = 64 MFMAs per wave, ONE barrier per step (slot 56: the
// step's last A read is issued at slot 52, the next step's first right behind the barrier, the producer's stores sit in
// slots 4-11); A fragments two positions ahead in a ring of four register quads (colours 0 1 2 x 4, then 0 1 2 3), B
// fragments straight from the transformed filters seven positions ahead in a ring of eight.  All 256 accumulators carry
// literal names a[16 q : 16 q + 15] inside inline-asm statements (see winograd3w.hip for why); tests/test_build_audit.py
// checks the generated code.
//
// SPLIT instantiation (small batches: the reference's 32 boxes per image).  A wave's K loop is 2048 MFMAs long whatever
// the batch, so a launch that cannot fill the chip takes ~76 us at ANY batch up to 128 (32 workgroups at B = 32).  There
// the channel range is cut into slices: workgroup (tile block, channel block, slice) runs K / slices channels and stores
// its PARTIAL 3x3 outputs (the output transform is linear) into its slice of a scratch tensor; a second small launch adds
// the slices in a fixed order, applies bias / ReLU and writes y -- deterministic, no atomics, no zero-fill.
#include <atomic>
#include <type_traits>
#include <utility>

#include "common.h"
#include "wino3_filter.h"
#include "wino3_transforms.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using mpsr::FastDiv;
using mpsr::fdiv;
using mpsr::w3t::at3z;
using mpsr::w3t::bt4z;

namespace f3z {
constexpr int KC = 8, NP = 16;
constexpr unsigned OOB = 0x80000000u;
constexpr int cA(int p) { return p < 12 ? p % 3 : p - 12; }  // ring colour of position p's A fragment (4 quads)
constexpr int BPRE = 7;   // B fragments requested this many positions ahead (28 MFMAs), ring of eight: 16 = 8 + 8
constexpr int BRING = 8;
constexpr int cB(int p) { return p % 8; }  // ... of its B fragment
}  // namespace f3z

struct Wino3ZParams {
    const float *x, *u, *bias, *mask;
    float *y;
    int H, W, C, N, dil, T;  // T = B * dil * dil tiles (one per pixel sub-grid)
    int cblocks, nblocks, mblocks, relu;
    unsigned xbytes, ubytes, ybytes;
    FastDiv div_tpi, div_d;  // tiles per image = dil^2, dil
    // SPLIT: K slices; slice k runs channel steps [k * steps, (k + 1) * steps) and stores into part + k * ybytes / 4
    float *part;
    int nslices, steps;
};

#ifdef W3Z_TRACE  // (timing builds only: tools/wino3w_trace.py) cycle stamps of the first eight workgroups' waves
__device__ unsigned long long g_w3z_trace[8 * 4 * 40];
#define W3Z_STAMP(i)                                                                                        \
    do {                                                                                                    \
        if (blockIdx.x < 8 && (threadIdx.x & 63) == 0)                                                      \
            g_w3z_trace[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 40 + (i)] = __builtin_readcyclecounter();  \
    } while (0)
#else
#define W3Z_STAMP(i) do { } while (0)
#endif

template <int V>
using ICZ = std::integral_constant<int, V>;
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f)
{
    (f(ICZ<Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// All sixteen positions accumulate in the accumulator half of the register file under LITERAL names (position q =
// a[16 q : 16 q + 15]); the names are the kernel's own by the clobber list of W3Z_CLAIM_ACC.  Why, and what the audit
// in tests/test_build_audit.py checks: winograd3w.hip.
// (no wait states inside the statements: a vector-ALU write of an A / B operand needs two before the MFMA that reads it,
// which hipcc does not pad for inline asm -- the operands here are written by LDS / buffer loads only, and
// tests/test_build_audit.py checks the compiled kernel for a vector-ALU write of an operand in the two instructions in
// front of each MFMA; a blanket s_nop 1 measured 1 % of the K loop)
#ifndef W3Z_BAUX
#define W3Z_BAUX 0  // cache policy of the transformed-filter loads: 2 = non-temporal (A/B builds)
#endif
#ifndef W3Z_STORE_AUX
#define W3Z_STORE_AUX 0  // cache policy of the result stores: 2 = non-temporal (A/B builds)
#endif
#ifdef W3Z_PAD_NOP
#define W3Z_PAD "s_nop 1\n\t"
#else
#define W3Z_PAD ""
#endif
#define W3Z_MFMA_A(q, a, b)                                                                                   \
    asm volatile(W3Z_PAD "v_mfma_f32_32x32x2_f32 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(16 * (q)), \
                 "i"(16 * (q) + 15))
#define W3Z_ZERO16(b)                                                                                                     \
    asm volatile("v_accvgpr_write_b32 a%c0, 0\n\tv_accvgpr_write_b32 a%c1, 0\n\tv_accvgpr_write_b32 a%c2, 0\n\t"          \
                 "v_accvgpr_write_b32 a%c3, 0\n\tv_accvgpr_write_b32 a%c4, 0\n\tv_accvgpr_write_b32 a%c5, 0\n\t"          \
                 "v_accvgpr_write_b32 a%c6, 0\n\tv_accvgpr_write_b32 a%c7, 0\n\tv_accvgpr_write_b32 a%c8, 0\n\t"          \
                 "v_accvgpr_write_b32 a%c9, 0\n\tv_accvgpr_write_b32 a%c10, 0\n\tv_accvgpr_write_b32 a%c11, 0\n\t"        \
                 "v_accvgpr_write_b32 a%c12, 0\n\tv_accvgpr_write_b32 a%c13, 0\n\tv_accvgpr_write_b32 a%c14, 0\n\t"       \
                 "v_accvgpr_write_b32 a%c15, 0" ::"i"((b)), "i"((b) + 1), "i"((b) + 2), "i"((b) + 3), "i"((b) + 4),       \
                 "i"((b) + 5), "i"((b) + 6), "i"((b) + 7), "i"((b) + 8), "i"((b) + 9), "i"((b) + 10), "i"((b) + 11),      \
                 "i"((b) + 12), "i"((b) + 13), "i"((b) + 14), "i"((b) + 15))
#define W3Z_READ_ACC(dst, idx) asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(dst) : "i"(idx))
#define W3Z_CLAIM_ACC()                                                                                                   \
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
template <bool MASK, bool SPLIT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino3z_conv_kernel(const Wino3ZParams p)
{
    static_assert(!(MASK && SPLIT), "the masked (training) launches are never split");
    using namespace f3z;
    constexpr int WM = 1, MT = 32, NT = 128;
    constexpr int APOS = MT * KC;    // floats per position of an A buffer
    constexpr int ABUF = NP * APOS;  // one A buffer (WM = 1: 25.6 KB)
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int xcd = blockIdx.x & 7, l_ = blockIdx.x >> 3;
    const int nb = l_ % p.nblocks;
    int l2_ = l_ / p.nblocks, slice = 0;
    if constexpr (SPLIT) {
        slice = l2_ % p.nslices;
        l2_ /= p.nslices;
    }
    const int mb = l2_ * 8 + xcd;
    if (mb >= p.mblocks) return;  // block-uniform
    const int n0 = nb * NT, t0 = mb * MT;
    W3Z_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wave % WM, ni = wave / WM;
    const int s0 = SPLIT ? slice * p.steps : 0;                 // first channel step of this workgroup
    const int send = SPLIT ? s0 + p.steps : p.cblocks, d = p.dil, tpi = d * d;

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
    float pa[16];  // the 4x4 transformed patch being built (one patch at a time): pa[4 u + v]
    auto vertical = [&](auto rc, auto jc) __attribute__((always_inline)) {  // data column j -> the four rows of it, kept in pa[4 u + j]
        constexpr int r = decltype(rc)::value, j = decltype(jc)::value;
        bt4z(raw[r][3 * j], raw[r][3 * j + 1], raw[r][3 * j + 2], pa[j], pa[4 + j], pa[8 + j], pa[12 + j]);
    };
    auto horizontal = [&](auto ic) __attribute__((always_inline)) {  // row u: its three column values -> four
        constexpr int i = decltype(ic)::value;
        float t0_, t1_, t2_, t3_;
        bt4z(pa[4 * i], pa[4 * i + 1], pa[4 * i + 2], t0_, t1_, t2_, t3_);
        pa[4 * i] = t0_;
        pa[4 * i + 1] = t1_;
        pa[4 * i + 2] = t2_;
        pa[4 * i + 3] = t3_;
    };
    auto store_a = [&](auto rc, int buf, int pos) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        awr[r][buf * ABUF + pos * APOS] = pa[pos];
    };
    // producer duty of slot m of a K step: the patch transformed in slots 0-11 (3 columns, then a row every other slot with its
    // four stores behind it), the next-but-one step's nine requests from slot 16 on
    auto duty = [&](int s, auto bufc, auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value, nbuf = decltype(bufc)::value ^ 1;
#ifdef W3Z_SKIP_PROD  // (timing experiments only: wrong results)
        return;
#endif
        static_for<WM>([&](auto rc) __attribute__((always_inline)) {
            constexpr int o = m;
            if constexpr (o >= 0 && o < 3) vertical(rc, ICZ<o>{});
            if constexpr (o >= 3 && o < 11 && ((o - 3) & 1) == 0) horizontal(ICZ<(o - 3) / 2>{});
            if constexpr (o >= 4 && o < 12) {
                constexpr int i = (o - 4) / 2;
                if constexpr (((o - 4) & 1) == 0) {
                    store_a(rc, nbuf, 4 * i);
                    store_a(rc, nbuf, 4 * i + 1);
                } else {
                    store_a(rc, nbuf, 4 * i + 2);
                    store_a(rc, nbuf, 4 * i + 3);
                }
            }
            constexpr int L = m - 16;
            if constexpr (L >= 0 && L < 9) load_raw(s + 2, rc, ICZ<L>{});
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
#ifdef W3Z_SKIP_BLOAD
        if (step > 0) return;
#endif
        const unsigned so = ((unsigned)step * (unsigned)NP + (unsigned)q) * bpstride;
        fb[cB(q)] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ru, bvoff, so, W3Z_BAUX));
    };
    // ---- A fragments: lane = (tile = lane & 31 of the wave's 32, k half = lane >> 5)
    const float *ard = lds + (32 * mi + (lane & 31)) * 8 + 4 * ((lane >> 5) ^ (((lane & 31) >> 3) & 1));
    float4 fa[4];
    auto load_a = [&](int buf, auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
#ifdef W3Z_SKIP_ALOAD
        if (buf >= 0 && q > 1) return;
#endif
        fa[cA(q)] = *reinterpret_cast<const float4 *>(ard + buf * ABUF + q * APOS);
    };

    auto mfma = [&](auto qc, float av, float bv) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        W3Z_MFMA_A(q, av, bv);
    };

    // ---- prologue: requests of step 0, the first B fragments, accumulators, A of step 0, requests of step 1
    static_for<WM>([&](auto rc) __attribute__((always_inline)) {
        static_for<9>([&](auto Lc) __attribute__((always_inline)) { load_raw(s0, rc, Lc); });
    });
    static_for<BPRE>([&](auto qc) __attribute__((always_inline)) { load_b(s0, qc); });
    W3Z_CLAIM_ACC();
    static_for<16>([&](auto qc) __attribute__((always_inline)) { W3Z_ZERO16(16 * decltype(qc)::value); });
    static_for<WM>([&](auto rc) __attribute__((always_inline)) {
        static_for<3>([&](auto jc) __attribute__((always_inline)) { vertical(rc, jc); });
        static_for<4>([&](auto ic) __attribute__((always_inline)) { horizontal(ic); });
#pragma unroll
        for (int i = 0; i < 16; ++i) store_a(rc, 0, i);
        static_for<9>([&](auto Lc) __attribute__((always_inline)) { load_raw(s0 + 1, rc, Lc); });
    });
    __syncthreads();
    load_a(0, ICZ<0>{});
    load_a(0, ICZ<1>{});
    W3Z_STAMP(1);

    // ---- K loop: slot m = MFMA k = m % 4 of position q = m / 4, followed by the slot's loads and producer work
    auto kstep = [&](int s, auto bufc) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        static_for<4 * NP>([&](auto mc) __attribute__((always_inline)) {
            constexpr int m = decltype(mc)::value, q = m / 4, k = m % 4;
            const float av = k == 0 ? fa[cA(q)].x : k == 1 ? fa[cA(q)].y : k == 2 ? fa[cA(q)].z : fa[cA(q)].w;
            const float bv = k == 0 ? fb[cB(q)].x : k == 1 ? fb[cB(q)].y : k == 2 ? fb[cB(q)].z : fb[cB(q)].w;
            mfma(ICZ<q>{}, av, bv);
            if constexpr (k == 0) {
#ifndef W3Z_NO_BARRIER
                if constexpr (m == 4 * (NP - 2)) __syncthreads();
#endif
                if constexpr (q + 2 < NP) load_a(buf, ICZ<(q + 2) % NP>{});
                else load_a(buf ^ 1, ICZ<(q + 2) % NP>{});
            }
            if constexpr (k == 1) {
                if constexpr (q + BPRE < NP) load_b(s, ICZ<(q + BPRE) % NP>{});
                else load_b(s + 1, ICZ<(q + BPRE) % NP>{});
            }
            duty(s, bufc, mc);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    for (int s = s0; s < send; s += 2) {
        kstep(s, ICZ<0>{});
        if (s - s0 < 32) W3Z_STAMP(2 + s - s0);
        kstep(s + 1, ICZ<1>{});
        if (s - s0 < 32) W3Z_STAMP(3 + s - s0);
    }
    // the last MFMAs' results: 18 wait states before anything reads them (hipcc pads nothing behind inline asm)
    W3Z_STAMP(34);
    int lane2 = lane;
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(lane2));

    // ---- epilogue, lane-local: register e of every position belongs to tile (e & 3) + 8 (e >> 2) + 4 (lane >> 5) of the
    // wave's 32 and to channel lane & 31: A^T M A on the lane's own 16 values, bias, ReLU / mask, nine strided pixels out
    const int n = n0 + 32 * ni + (lane2 & 31);
    // (SPLIT: the slice's partial outputs, no bias, no activation: wino3z_finish_kernel)
    const float bias_v = (!SPLIT && p.bias && n < p.N) ? p.bias[n] : 0.f;
    const float floor_v = (!SPLIT && p.relu) ? 0.f : -__builtin_inff();  // ReLU as one max
    float *ydst = SPLIT ? p.part + (size_t)slice * (p.ybytes / 4) : p.y;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(ydst, 0, (int)p.ybytes, 0x00020000);
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
        float z[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float mv[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) W3Z_READ_ACC(mv[v], 16 * (4 * u + v) + e);
            at3z(mv[0], mv[1], mv[2], mv[3], z[u][0], z[u][1], z[u][2]);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float yv[3];
            at3z(z[0][j], z[1][j], z[2][j], z[3][j], yv[0], yv[1], yv[2]);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float v = fmaxf(yv[i] + bias_v, floor_v);
                if constexpr (MASK) v = mk[3 * j + i] > 0.f ? v : 0.f;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ry, voff, so[3 * j + i], W3Z_STORE_AUX);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // (one register column at a time: hipcc otherwise hoists all 256 reads)
    });
    W3Z_STAMP(35);
}

}  // namespace

namespace {
// the filter transform of this form as a launch of its own (when no cache / tail job made it)
__global__ __launch_bounds__(256) void wino3z_filter_kernel(const float *__restrict__ w, int N, int C, float *__restrict__ u)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < (long long)N * C) mpsr::wino3z_filter_one(w, N, C, u, i);
}
}  // namespace

namespace {
// y = act(bias + sum of the K slices' partial outputs), slices added in index order; 16 bytes per thread
__global__ __launch_bounds__(256) void wino3z_finish_kernel(const float *__restrict__ part, const float *__restrict__ bias,
                                                            float *__restrict__ y, long long total4, long long slice4,
                                                            int nslices, int N4, int relu)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    float4 a = reinterpret_cast<const float4 *>(part)[i];
    for (int k = 1; k < nslices; ++k) {
        const float4 b = reinterpret_cast<const float4 *>(part)[i + k * slice4];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (bias) {
        const float4 b = reinterpret_cast<const float4 *>(bias)[i % N4];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (relu) a = make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
    reinterpret_cast<float4 *>(y)[i] = a;
}
}  // namespace

namespace mpsr {

static std::atomic<int> g_w3z_split{-1};  // mpsr_debug_set_wino3z_split: -1 by launch size, 0 never, k > 1 = k slices

// K slices of a launch: enough workgroups for every CU (up to `want_blocks`), at least `min_steps` channel steps each
// and an even number of them, a power of two that divides the steps; 1 = not split.  (Depends on the batch only through
// the workgroup count rounded up to whole XCD rounds.)  Shared with the F(2x2,3x3) kernel (winograd.hip).
int winograd_slices(long long blocks, int want_blocks, int cblocks, int min_steps, size_t part_floats, size_t y_floats)
{
    const int forced = g_w3z_split.load();
    if (forced == 0 || (y_floats & 3)) return 1;
    int s = 1;
    if (forced > 1) s = forced;
    else
        while (blocks * s * 2 <= want_blocks && s < 16) s *= 2;
    while (s > 1 && (cblocks % (2 * s) != 0 || cblocks / s < min_steps || (size_t)s * y_floats > part_floats)) s /= 2;
    return s;
}

int winograd_finish_slices(const float *part, const float *bias, float *y, size_t y_floats, int nslices, int N, int relu,
                           hipStream_t s)
{
    const long long total4 = (long long)(y_floats / 4);
    hipLaunchKernelGGL(wino3z_finish_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, part, bias, y, total4,
                       total4, nslices, N / 4, relu);
    MPSR_CHECK_LAUNCH("wino3z_finish_kernel");
    return MPSR_OK;
}

// launches the kernel on filters already transformed by wino3z_filter_one (wino3_filter.h: U[cb][16 positions][n][8])
int launch_winograd3z(const float *x, int B, int H, int W, int C, const float *u, const float *bias, int relu, float *y,
                      int N, int dilation, hipStream_t s, const float *mask, float *part, size_t part_floats)
{
    using namespace f3z;
    Wino3ZParams p;
    p.x = x; p.u = u; p.bias = bias; p.mask = mask; p.y = y;
    p.H = H; p.W = W; p.C = C; p.N = N; p.dil = dilation;
    p.T = B * dilation * dilation;
    p.cblocks = C / KC;
    constexpr int MT = 32, NT = 128;
    p.nblocks = ceil_div(N, NT);
    p.mblocks = ceil_div(p.T, MT);
    p.relu = relu;
    p.xbytes = (unsigned)((long long)B * H * W * C * 4);
    p.ubytes = (unsigned)((size_t)NP * N * C * 4);
    p.ybytes = (unsigned)((long long)B * H * W * N * 4);
    p.div_tpi = make_fastdiv(dilation * dilation);
    p.div_d = make_fastdiv(dilation);
    long long blocks = 8LL * ceil_div(p.mblocks, 8) * p.nblocks;
    if (blocks > 0x7fffffffLL) return fail(MPSR_ERR_UNSUPPORTED, "conv3x3_winograd3: grid too large");
    const size_t y_floats = (size_t)B * H * W * N;
    const bool splittable = part && N % 4 == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)part & 15) == 0 &&
                            (!bias || ((uintptr_t)bias & 15) == 0);
    p.nslices = mask ? 1 : winograd_slices(blocks, 256, p.cblocks, 4, splittable ? part_floats : 0, y_floats);
    p.steps = p.cblocks / p.nslices;
    p.part = part;
    const size_t ldsb = (size_t)2 * NP * MT * KC * sizeof(float);
    const void *kern = mask ? reinterpret_cast<const void *>(wino3z_conv_kernel<true>)
                            : p.nslices > 1 ? reinterpret_cast<const void *>(wino3z_conv_kernel<false, true>)
                                            : reinterpret_cast<const void *>(wino3z_conv_kernel<false>);
    MPSR_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    if (mask) hipLaunchKernelGGL((wino3z_conv_kernel<true>), dim3((unsigned)blocks), dim3(256), ldsb, s, p);
    else if (p.nslices > 1) {
        blocks *= p.nslices;
        hipLaunchKernelGGL((wino3z_conv_kernel<false, true>), dim3((unsigned)blocks), dim3(256), ldsb, s, p);
        MPSR_CHECK_LAUNCH("wino3z_conv_kernel");
        return winograd_finish_slices(part, bias, y, y_floats, p.nslices, N, relu, s);
    } else hipLaunchKernelGGL((wino3z_conv_kernel<false>), dim3((unsigned)blocks), dim3(256), ldsb, s, p);
    MPSR_CHECK_LAUNCH("wino3z_conv_kernel");
    return MPSR_OK;
}

int launch_winograd3z_filter(const float *w, int N, int C, float *u, hipStream_t s)
{
    const long long total = (long long)N * C;
    hipLaunchKernelGGL(wino3z_filter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, N, C, u);
    MPSR_CHECK_LAUNCH("wino3z_filter_kernel");
    return MPSR_OK;
}

}  // namespace mpsr

extern "C" void mpsr_debug_set_wino3z_split(int slices) { mpsr::g_w3z_split = slices; }

#ifdef W3Z_TRACE
extern "C" int mpsr_debug_wino3z_trace(unsigned long long *host_out, int count)
{
    if (count > 8 * 4 * 40) count = 8 * 4 * 40;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_w3z_trace), sizeof(unsigned long long) * count) == hipSuccess ? 0 : 1;
}
#endif
