// Weight gradient of the atrous 3x3 layers whose pixel sub-grids are single 3x3 tiles (ResNet-101 block3's conv2 at output
// stride 4: 12x12 maps at dilation 4, 23 layers per training step; reference: TensorFlow autodiff of
// object_detection/nets/resnet_v1.py:116-127 under monopsr/core/trainer.py:71-81) in the transform domain of the
// SIXTEEN-product form of a zero-padded tile (wino3_transforms.h; forward: winograd3z.hip).
//
//   forward    Y (3x3) = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A                (4x4 element positions)
//   gradient   dg = G^T [ sum over tiles of (A dY A^T) (.) (B^T d B) ] G
// i.e. per element position p (16 of them) one GEMM over the tiles t (one tile per pixel sub-grid: B x dilation^2 of them)
//   dU_p[n][c] = sum_t Yh_p[t][n] * V_p[t][c],        Yh = A dY A^T (3x3 -> 4x4),  V = B^T d B (the forward's transform)
// 16 products per (channel pair, tile) where the border-class direct weight gradient (backward.hip) executes 49: a third
// of the matrix work of 23 x 237-258 us of the training step.  Both operands are transformed on the fly, neither reaches
// HBM; the fold back G^T dU G is lane-local (below) and the tile slices meet in dw itself through fp32 atomics.
//
// Kernel form = winograd3z.hip's: ONE WAVE OWNS ALL 16 POSITIONS of a 32 (n) x 32 (c) block -- 256 accumulator registers
// under literal names in the accumulator half of the register file, one wave per SIMD -- so a workgroup (4 waves = 64 n x
// 64 c) transforms each operand patch ONCE for two blocks: 2 patches per thread and K step of 4 tiles for 32 MFMAs per
// wave.  Operands in LDS as [stage][Yh | V][position][row][4 tiles], the tiles of a row ordered (0, 2, 1, 3): lane (row,
// k half h) reads the 8 bytes at 8 h = its k of both MFMAs of the step; double buffered, one barrier per step.  The patches
// of the step after next are requested in the first slots of a step into the OTHER of two register sets: a full step of
// lead (the vector half of the register file is nearly empty here).
#include <type_traits>
#include <utility>

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"
#include "wino3_transforms.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using mpsr::FastDiv;
using mpsr::fdiv;
using mpsr::w3t::a4z;
using mpsr::w3t::bt4z;

namespace w3g {
constexpr int KT = 4, NP = 16;
constexpr int POSF = 64 * KT;        // floats per position of one operand (64 rows x 4 tiles)
constexpr int OPF = NP * POSF;       // one operand of a stage (4096 floats = 16 KB)
constexpr int STAGEF = 2 * OPF;      // Yh then V
constexpr int LDSF = 2 * STAGEF;     // 16384 floats = 64 KB
constexpr unsigned OOB = 0x80000000u;
constexpr int cF(int p) { return p < 12 ? p % 3 : p - 12; }  // ring colour of position p's fragments (4 pairs)
}  // namespace w3g

struct W3gParams {
    const float *x, *dy;
    float *dw;  // (N, 9 C): accumulated into
    float *db;  // (N) or nullptr: the bias gradient, accumulated into (column sums of dY)
    int H, W, C, N, dil, T;
    int nblocks, cblocks, nslices, steps;  // steps: K steps (of 4 tiles) per slice, even
    FastDiv div_tpi, div_d;
    unsigned xbytes, dybytes;
};

template <int V>
using ICG = std::integral_constant<int, V>;
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f)
{
    (f(ICG<Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// (literal accumulator-register names, no wait states inside the statements: see winograd3w.hip; the same audit in
// tests/test_build_audit.py covers this kernel)
#define W3G_MFMA_A(q, a, b)                                                                                 \
    asm volatile("v_mfma_f32_32x32x2_f32 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(16 * (q)), \
                 "i"(16 * (q) + 15))
#define W3G_ZERO16(b)                                                                                                     \
    asm volatile("v_accvgpr_write_b32 a%c0, 0\n\tv_accvgpr_write_b32 a%c1, 0\n\tv_accvgpr_write_b32 a%c2, 0\n\t"          \
                 "v_accvgpr_write_b32 a%c3, 0\n\tv_accvgpr_write_b32 a%c4, 0\n\tv_accvgpr_write_b32 a%c5, 0\n\t"          \
                 "v_accvgpr_write_b32 a%c6, 0\n\tv_accvgpr_write_b32 a%c7, 0\n\tv_accvgpr_write_b32 a%c8, 0\n\t"          \
                 "v_accvgpr_write_b32 a%c9, 0\n\tv_accvgpr_write_b32 a%c10, 0\n\tv_accvgpr_write_b32 a%c11, 0\n\t"        \
                 "v_accvgpr_write_b32 a%c12, 0\n\tv_accvgpr_write_b32 a%c13, 0\n\tv_accvgpr_write_b32 a%c14, 0\n\t"       \
                 "v_accvgpr_write_b32 a%c15, 0" ::"i"((b)), "i"((b) + 1), "i"((b) + 2), "i"((b) + 3), "i"((b) + 4),       \
                 "i"((b) + 5), "i"((b) + 6), "i"((b) + 7), "i"((b) + 8), "i"((b) + 9), "i"((b) + 10), "i"((b) + 11),      \
                 "i"((b) + 12), "i"((b) + 13), "i"((b) + 14), "i"((b) + 15))
#define W3G_READ_ACC(dst, idx) asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(dst) : "i"(idx))
#define W3G_CLAIM_ACC()                                                                                                   \
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

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino3_wgrad_kernel(const W3gParams p)
{
    using namespace w3g;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // XCD x (workgroup b runs on XCD b % 8: speed only) takes the tile slices x, x + 8, ...; the (n, c) blocks of one slice
    // are consecutive workgroups of it, so the slice's pixels come from HBM once
    const int xcd = blockIdx.x & 7, l_ = blockIdx.x >> 3;
    const int nblk = p.nblocks * p.cblocks;
    const int blk = l_ % nblk, slice = (l_ / nblk) * 8 + xcd;
    if (slice >= p.nslices) return;  // block-uniform
    const int n0 = (blk / p.cblocks) * 64, c0 = (blk % p.cblocks) * 64;
    const int t_begin = slice * p.steps * KT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wave >> 1, ni = wave & 1;  // the wave's 32-row block of n (Yh rows) and of c (V rows)
    const int d = p.dil, tpi = d * d;

    // ---- producers: thread = (tile lt of the step, operand row pr) for ONE dY block and ONE x patch per step; a wave covers
    // 4 tiles x 16 rows, so its 64 stores of a position are 64 consecutive floats
    const int lt = lane >> 4, pr = 16 * wave + (lane & 15);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.dy), 0, (int)p.dybytes, 0x00020000);
    float rawy[2][9], rawx[2][9];  // [step parity][3 j + i] = sub-grid pixel (row i, column j) of the step's dY block / x patch
    // byte offsets of this thread's dY block / x patch of K step `step` (out of range past the slice's last step or the last
    // tile: zeros), computed once per step; then the nine requests of each
    unsigned voy = 0, vox = 0;
    auto locate = [&](int step) __attribute__((always_inline)) {
        const int t = t_begin + step * KT + lt;
        const int img = fdiv(t, p.div_tpi), sub = t - img * tpi;
        const int a = fdiv(sub, p.div_d), b = sub - a * d;
        const bool in = t < p.T && step < p.steps;
        const int pix = (img * p.H + a) * p.W + b;
        voy = in ? (unsigned)(pix * p.N + n0 + pr) * 4u : OOB;
        vox = in ? (unsigned)(pix * p.C + c0 + pr) * 4u : OOB;
    };
    auto request = [&](auto setc, auto Lc, auto xc) __attribute__((always_inline)) {
        constexpr int set = decltype(setc)::value, L = decltype(Lc)::value, j = L / 3, i = L % 3;
        constexpr bool X = decltype(xc)::value != 0;
        if constexpr (X)
            rawx[set][L] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                         rx, vox, (unsigned)((d * i * p.W + d * j) * p.C) * 4u, 0));
        else
            rawy[set][L] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                         rdy, voy, (unsigned)((d * i * p.W + d * j) * p.N) * 4u, 0));
    };
    float pa[16];  // the 4x4 transformed block being built (one at a time): pa[4 u + v]
    // the bias gradient rides along: sum of the tile's nine dY values = w^T (A dY A^T) w with w = (0, 0, 5/6, -1/6) (rows 2, 3
    // of A are S +- 3/2 d1: (o2 - o3) / 2 + (o2 + o3) / 3 = d0 + d1 + d2); only the workgroups of the first c block add it
    float bsum = 0.f;
    const bool want_db = p.db != nullptr && c0 == 0;
    // [stage][operand][position][row][slot of the tile: tiles 0, 2, 1, 3]
    float *wr = lds + pr * KT + ((lt & 1) * 2 + (lt >> 1));
    auto store = [&](int stage, auto opc, int pos) __attribute__((always_inline)) {
        constexpr int op = decltype(opc)::value;
        wr[stage * STAGEF + op * OPF + pos * POSF] = pa[pos];
    };
    // duties of slot m of a K step whose stage is `st` (= the step's parity): the dY block of the NEXT step (register set
    // st ^ 1) in slots 0..11, its x patch in 12..23 (3 columns, then a row every other slot with its four stores behind it),
    // both into stage st ^ 1; the 18 requests of the step after next into register set st, one per slot from slot 0 on
    auto duty = [&](int s, auto stc, auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value, st = decltype(stc)::value, nst = st ^ 1;
        static_for<2>([&](auto opc) __attribute__((always_inline)) {
            constexpr int op = decltype(opc)::value, o = m - 12 * op;
            if constexpr (o >= 0 && o < 3) {
                if constexpr (op == 0) a4z(rawy[nst][3 * o], rawy[nst][3 * o + 1], rawy[nst][3 * o + 2], pa[o], pa[4 + o], pa[8 + o], pa[12 + o]);
                else bt4z(rawx[nst][3 * o], rawx[nst][3 * o + 1], rawx[nst][3 * o + 2], pa[o], pa[4 + o], pa[8 + o], pa[12 + o]);
            }
            if constexpr (o >= 3 && o < 11 && ((o - 3) & 1) == 0) {
                constexpr int i = (o - 3) / 2;
                float t0_, t1_, t2_, t3_;
                if constexpr (op == 0) a4z(pa[4 * i], pa[4 * i + 1], pa[4 * i + 2], t0_, t1_, t2_, t3_);
                else bt4z(pa[4 * i], pa[4 * i + 1], pa[4 * i + 2], t0_, t1_, t2_, t3_);
                pa[4 * i] = t0_;
                pa[4 * i + 1] = t1_;
                pa[4 * i + 2] = t2_;
                pa[4 * i + 3] = t3_;
                if constexpr (op == 0 && i == 2) bsum = fmaf(25.f / 36.f, t2_, fmaf(-5.f / 36.f, t3_, bsum));
                if constexpr (op == 0 && i == 3) bsum = fmaf(-5.f / 36.f, t2_, fmaf(1.f / 36.f, t3_, bsum));
            }
            if constexpr (o >= 4 && o < 12) {
                constexpr int i = (o - 4) / 2;
                if constexpr (((o - 4) & 1) == 0) {
                    store(nst, opc, 4 * i);
                    store(nst, opc, 4 * i + 1);
                } else {
                    store(nst, opc, 4 * i + 2);
                    store(nst, opc, 4 * i + 3);
                }
            }
        });
        if constexpr (m == 0) locate(s + 2);
        if constexpr (m < 9) request(stc, ICG<m>{}, ICG<0>{});
        if constexpr (m >= 9 && m < 18) request(stc, ICG<m - 9>{}, ICG<1>{});
    };

    // ---- fragments: lane (row = lane & 31 of the wave's block, k half h = lane >> 5) reads tiles (h, 2 + h) as 8 bytes
    const float *rdy_ = lds + (32 * mi + (lane & 31)) * KT + 2 * (lane >> 5);
    const float *rdv_ = lds + OPF + (32 * ni + (lane & 31)) * KT + 2 * (lane >> 5);
    f32x2 fa[4], fb[4];
    auto load_f = [&](int stage, auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        fa[cF(q)] = *reinterpret_cast<const f32x2 *>(rdy_ + stage * STAGEF + q * POSF);
        fb[cF(q)] = *reinterpret_cast<const f32x2 *>(rdv_ + stage * STAGEF + q * POSF);
    };

    auto mfma = [&](auto qc, float av, float bv) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        W3G_MFMA_A(q, av, bv);
    };

    // ---- prologue: patches of steps 0 and 1 requested, accumulators cleared, step 0 transformed into stage 0
    locate(0);
    static_for<9>([&](auto Lc) __attribute__((always_inline)) { request(ICG<0>{}, Lc, ICG<0>{}); });
    static_for<9>([&](auto Lc) __attribute__((always_inline)) { request(ICG<0>{}, Lc, ICG<1>{}); });
    locate(1);
    static_for<9>([&](auto Lc) __attribute__((always_inline)) { request(ICG<1>{}, Lc, ICG<0>{}); });
    static_for<9>([&](auto Lc) __attribute__((always_inline)) { request(ICG<1>{}, Lc, ICG<1>{}); });
    W3G_CLAIM_ACC();
    static_for<16>([&](auto qc) __attribute__((always_inline)) { W3G_ZERO16(16 * decltype(qc)::value); });
    {   // (the transform half of the duties of a step "-1" with stage 1: register set 0 into stage 0)
        static_for<2>([&](auto opc) __attribute__((always_inline)) {
            constexpr int op = decltype(opc)::value;
            static_for<3>([&](auto jc) __attribute__((always_inline)) {
                constexpr int o = decltype(jc)::value;
                if constexpr (op == 0) a4z(rawy[0][3 * o], rawy[0][3 * o + 1], rawy[0][3 * o + 2], pa[o], pa[4 + o], pa[8 + o], pa[12 + o]);
                else bt4z(rawx[0][3 * o], rawx[0][3 * o + 1], rawx[0][3 * o + 2], pa[o], pa[4 + o], pa[8 + o], pa[12 + o]);
            });
            static_for<4>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                float t0_, t1_, t2_, t3_;
                if constexpr (op == 0) a4z(pa[4 * i], pa[4 * i + 1], pa[4 * i + 2], t0_, t1_, t2_, t3_);
                else bt4z(pa[4 * i], pa[4 * i + 1], pa[4 * i + 2], t0_, t1_, t2_, t3_);
                pa[4 * i] = t0_;
                pa[4 * i + 1] = t1_;
                pa[4 * i + 2] = t2_;
                pa[4 * i + 3] = t3_;
                if constexpr (op == 0 && i == 2) bsum = fmaf(25.f / 36.f, t2_, fmaf(-5.f / 36.f, t3_, bsum));
                if constexpr (op == 0 && i == 3) bsum = fmaf(-5.f / 36.f, t2_, fmaf(1.f / 36.f, t3_, bsum));
            });
#pragma unroll
            for (int i = 0; i < 16; ++i) store(0, opc, i);
        });
    }
    __syncthreads();
    load_f(0, ICG<0>{});
    load_f(0, ICG<1>{});

    // ---- K loop: slot m = MFMA j = m % 2 of position q = m / 2
    auto kstep = [&](int s, auto stc) __attribute__((always_inline)) {
        constexpr int st = decltype(stc)::value;
        static_for<2 * NP>([&](auto mc) __attribute__((always_inline)) {
            constexpr int m = decltype(mc)::value, q = m / 2, j = m % 2;
            const float av = j == 0 ? fa[cF(q)].x : fa[cF(q)].y;
            const float bv = j == 0 ? fb[cF(q)].x : fb[cF(q)].y;
            mfma(ICG<q>{}, av, bv);
            if constexpr (j == 0) {
                if constexpr (m == 2 * (NP - 2)) __syncthreads();
                if constexpr (q + 2 < NP) load_f(st, ICG<(q + 2) % NP>{});
                else load_f(st ^ 1, ICG<(q + 2) % NP>{});
            }
            duty(s, stc, mc);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    for (int s = 0; s < p.steps; s += 2) {
        kstep(s, ICG<0>{});
        kstep(s + 1, ICG<1>{});
    }
    int lane2 = lane;
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(lane2));

    // ---- bias gradient: the four tile-lanes of a row (lane, lane + 16, + 32, + 48) meet in the first, one atomic per row
    if (want_db) {  // block-uniform
        float b = bsum;
        b += __shfl_xor(b, 16);
        b += __shfl_xor(b, 32);
        if ((lane2 >> 4) == 0 && n0 + pr < p.N) unsafeAtomicAdd(&p.db[n0 + pr], b);
    }

    // ---- epilogue, lane-local: accumulator element e of every position belongs to row n = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    // of the wave's block and column c = lane & 31, so the fold back to the 3x3 filter, G^T dU G (G = the forward's 4x3:
    // wino3_filter.h), is register arithmetic on the lane's own 16 values -- and the 9 results go straight into dw with fp32
    // atomics (16 tile slices meet there): 144 atomics per lane, no scratch, no memset, no second kernel
    const int cc = c0 + 32 * ni + (lane2 & 31);
    auto gt = [](float m0, float m1, float m2, float m3, float &o0, float &o1, float &o2) __attribute__((always_inline)) {
        const float d01 = m0 - m1, s01 = m0 + m1, d23 = m2 - m3, s23 = m2 + m3;  // G^T (3x4) applied to a 4-vector
        const float t = d23 * (1.f / 6.f);
        o0 = fmaf(-0.25f, d01, t);
        o1 = fmaf(0.125f, s01, 0.25f * s23);
        o2 = fmaf(0.25f, d01, t);
    };
    static_for<16>([&](auto ec) __attribute__((always_inline)) {
        constexpr int e = decltype(ec)::value;
        float t[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // over the rows u of column j: positions 4 u + j
            float col[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) W3G_READ_ACC(col[r], 16 * (4 * r + j) + e);
            gt(col[0], col[1], col[2], col[3], t[0][j], t[1][j], t[2][j]);
        }
        const int n = n0 + 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * (lane2 >> 5);
        float *dst = p.dw + (size_t)n * 9 * p.C + cc;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float o[3];
            gt(t[a][0], t[a][1], t[a][2], t[a][3], o[0], o[1], o[2]);
#pragma unroll
            for (int b = 0; b < 3; ++b) {
#ifdef W3G_NO_ATOMICS  // (timing experiment only: wrong results)
                if (o[b] == 123.456f) dst[0] = o[b];
#else
                if (n < p.N && cc < p.C) unsafeAtomicAdd(dst + (size_t)(a * 3 + b) * p.C, o[b]);
#endif
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

}  // namespace

namespace mpsr {

// Shapes the F(3x3,3x3) weight gradient takes: 3x3, one tile per pixel sub-grid (H = W = 3 x dilation), 64-channel blocks,
// enough tiles to give every CU a workgroup a few hundred steps long, 32-bit byte offsets.
bool winograd3_wgrad_applies(int B, int H, int W, int C, int N, int KH, int KW, int dilation)
{
    const long long M = (long long)B * H * W;
    return KH == 3 && KW == 3 && dilation >= 1 && H == W && H == 3 * dilation && C % 64 == 0 && N % 64 == 0 && C >= 128 &&
           N >= 128 && (long long)B * dilation * dilation >= 2048 && M * C * 4 < 0x7f000000LL && M * N * 4 < 0x7f000000LL;
}
// (no scratch: the kernel accumulates into dw directly)
int conv3x3_wgrad_winograd3(const float *x, const float *dy, int B, int H, int W, int C, int N, int dilation, float *dw,
                            float *db, hipStream_t s)
{
    using namespace w3g;
    MPSR_REQUIRE(winograd3_wgrad_applies(B, H, W, C, N, 3, 3, dilation), "conv3x3_wgrad_winograd3: unsupported shape");
    W3gParams p;
    p.x = x; p.dy = dy; p.dw = dw; p.db = db;
    p.H = H; p.W = W; p.C = C; p.N = N; p.dil = dilation;
    p.T = B * dilation * dilation;
    p.nblocks = N / 64; p.cblocks = C / 64;
    p.div_tpi = make_fastdiv(dilation * dilation);
    p.div_d = make_fastdiv(dilation);
    p.xbytes = (unsigned)((long long)B * H * W * C * 4);
    p.dybytes = (unsigned)((long long)B * H * W * N * 4);
    // one workgroup per CU (one 400-accumulator wave per SIMD): one round of slices, each an even number of 4-tile steps
    const int blocks = p.nblocks * p.cblocks;
    int slices = (256 / blocks + 7) / 8 * 8;  // (a multiple of 8: slice i runs on XCD i % 8)
    if (slices < 8) slices = 8;
    int steps = (int)(((long long)p.T + (long long)slices * KT - 1) / ((long long)slices * KT));
    steps = (steps + 1) / 2 * 2;
    p.steps = steps;
    p.nslices = (int)(((long long)p.T + (long long)steps * KT - 1) / ((long long)steps * KT));
    MPSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wino3_wgrad_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDSF * sizeof(float))));
    const unsigned grid = (unsigned)(((p.nslices + 7) / 8) * 8 * blocks);
    hipLaunchKernelGGL(wino3_wgrad_kernel, dim3(grid), dim3(256), LDSF * sizeof(float), s, p);
    MPSR_CHECK_LAUNCH("wino3_wgrad_kernel");
    return MPSR_OK;
}

}  // namespace mpsr
