// Stride-1 SAME (optionally atrous) convolution / fully-connected layer as an fp32-MFMA implicit GEMM with a fused
// bias + residual + ReLU epilogue, for gfx950.
//
//   y[m][n] = act( sum_{tap,c} x[pixel(m) + tap][c] * w[n][tap*C + c] + bias[n] + residual[m][n] )
//   m = output pixel (b,y,x) in NHWC order, n = output channel, K = KH*KW*C.
//
// Why fp32 MFMA: the path must match an fp32 TensorFlow graph to 1e-3 through ~100 sequential layers;
// v_mfma_f32_32x32x2_f32 is exact fp32 (an fmaf chain) at the fp32 peak of 157 TFLOP/s, 2.4x what a VALU GEMM
// reaches (cdna_hip_programming.md 3), and leaves the VALU free for the im2col address arithmetic.
//
// Structure (one workgroup = 256 threads = 4 waves, BM x BN output tile, BK = 32):
//   * A tile (BM pixels x 32 channels of one tap) is gathered straight from the NHWC activation: every thread owns
//     fixed pixel rows for the whole K loop, so a tap is one add + bounds test per row.  B tile (BN filters x 32)
//     comes from the (N, K) weight matrix.  16-byte buffer loads with hardware bounds checking: out-of-image taps,
//     rows past M / N and the K tail return zeros for one v_cndmask on the offset, so the K loop is one basic block.
//   * Both tiles sit in LDS as [row][32 + 4 pad] floats: the 144-byte row stride makes the ds_read_b128 fragment
//     reads and the ds_write_b128 fills bank-conflict free (MI355X_MICROARCH.md LDS table).
//   * K ordering inside a tile is chosen so a lane's four consecutive floats feed four consecutive MFMAs: lane
//     (i = l & 31, h = l >> 5) reads A[i][8*kb + 4*h .. +3]; MFMA step s multiplies k = 8*kb + 4*h + s.
//   * Software pipeline: next tile's global loads are issued into registers before the current tile's MFMAs and
//     written to LDS after them (one LDS buffer, two barriers per K step); 2-7 workgroups per CU overlap.
//   * Workgroup ids are remapped so consecutive ids of one XCD walk the N tiles of one M panel: the activation
//     panel is fetched from HBM once and re-read from that XCD's L2.
//   * Border-class tiling for atrous 3x3 layers: with dilation d on an H x W map, a pixel in the first / last d rows
//     (columns) has one row (column) of taps entirely outside the image.  Output pixels are therefore grouped into
//     up to 9 rectangles ("classes") whose pixels share the same set of in-image taps; an M tile holds pixels of ONE
//     class (from several images) and its K loop visits only that class's taps.  On the 12x12, d = 4 block3 layers
//     this skips 40 % of the multiply-adds (4, 6 or 9 taps instead of 9), on block2 (d = 2) 21 % -- with results
//     bit-identical to visiting the zero taps.
//   * split_k > 1 writes raw partial tiles to a workspace; a second kernel reduces and applies the epilogue
//     (used for the K = 18432 fully-connected layers where M = batch is small).
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// Arithmetic of the contraction.  MATH_FP32 (default): v_mfma_f32_32x32x2_f32, exact fp32 products.
// MATH_BF16X3 (opt-in, mpsr_set_conv_math): every fp32 operand is split into hi + lo bfloat16 halves when its tile is
// written to LDS (hi = rne(x), lo = rne(x - hi): 16 mantissa bits together) and each product is evaluated as
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- 3 matrix instructions at 16x the fp32
// MFMA rate.  Per-product relative error <= ~2^-16; measured end-to-end drift of the whole network 1e-5..4e-5
// (DESIGN.md 4.1), against the path's 1e-3 budget.  Same tiles, loads, LDS footprint and epilogue as the fp32 path.
constexpr int MATH_FP32 = 0, MATH_BF16X3 = 1;

constexpr int BK = 32;
constexpr int LDS_STRIDE = BK + 4;  // floats per LDS row
constexpr int MAX_CLS = 9;

// A rectangle of output pixels [y0, y0+h) x [x0, x0+w) of every image whose in-image taps are ky0..ky1 x kx0..kx1.
struct PixelClass {
    int y0, h, x0, w;
    int ky0, ky1, kx0, kx1;
    int tiles;  // M tiles of this class
    int rows;   // B * h * w
};

struct ConvParams {
    const float *x;
    const float *w;
    const float *bias;
    const float *residual;
    float *y;
    float *ws;
    int M, H, W, C, N, KH, KW, dil, relu;
    int ksteps_total, ksteps_per_split, cblocks;
    int mtiles_xcd, ntiles, splits;  // mtiles_xcd: M tiles per XCD (max over XCDs)
    unsigned xbytes, wbytes;
    int ncls;
    PixelClass cls[MAX_CLS];
};

template <int BM, int BN, int WM, int WN, int MATH = MATH_FP32>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p)
{
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int AV = BM / 32, BV = BN / 32;  // float4 loads per thread per tile
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDS_STRIDE];
    float *As = lds, *Bs = lds + BM * LDS_STRIDE;

    // Tile order.  Workgroup b is observed to run on XCD b % 8 (speed only, never correctness): XCD x takes M tiles
    // x, x+8, x+16, ... of EVERY pixel class -- so each XCD gets the same mix of long and short K loops -- and walks
    // them heaviest class first, the N tiles of one M tile back to back so its activation panel stays in that
    // XCD's L2.  Workgroups past an XCD's share (classes whose tile count is not a multiple of 8) exit at once.
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int ni = l % p.ntiles;
    int lm = (l / p.ntiles) % p.mtiles_xcd;
    const int si = l / (p.ntiles * p.mtiles_xcd);
    const int n0 = ni * BN;
    int ci = -1;
    for (int c = 0; c < p.ncls; ++c) {
        const int cnt = p.cls[c].tiles > xcd ? (p.cls[c].tiles - xcd + 7) >> 3 : 0;
        if (lm < cnt) {
            ci = c;
            break;
        }
        lm -= cnt;
    }
    if (ci < 0) return;  // block-uniform
    const PixelClass pc = p.cls[ci];
    const int r0 = (xcd + 8 * lm) * BM;  // first row of the tile inside its class
    const int ppi = pc.h * pc.w;               // class pixels per image
    const int ntx = pc.kx1 - pc.kx0 + 1;
    const int ksteps_cls = (pc.ky1 - pc.ky0 + 1) * ntx * p.cblocks;
    // split-K: every class shares ITS K steps among the p.splits slices (an empty slice writes a zero partial)
    const int ks_per = p.splits > 1 ? (ksteps_cls + p.splits - 1) / p.splits : ksteps_cls;
    const int ks_begin = si * ks_per;
    const int ks_end = min(ksteps_cls, ks_begin + ks_per);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = tid >> 3, lcol = (tid & 7) * 4;

    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, (int)p.wbytes, 0x00020000);

    // class row -> (y, x, linear NHWC pixel index); -1 for rows past the end of the class
    auto decode = [&](int row, int &yy, int &xx) -> int {
        if (row >= pc.rows) {
            yy = -(1 << 20);  // fails every bounds test
            xx = 0;
            return -1;
        }
        const int img = row / ppi, pp = row - img * ppi;
        const int py = pp / pc.w;
        yy = pc.y0 + py;
        xx = pc.x0 + (pp - py * pc.w);
        return (img * p.H + yy) * p.W + xx;
    };

    // per-thread A rows: pixel coordinates and byte offset are fixed for the whole K loop
    int ay[AV], ax[AV];
    unsigned abase[AV];
#pragma unroll
    for (int j = 0; j < AV; ++j) {
        const int pix = decode(r0 + lrow + 32 * j, ay[j], ax[j]);
        abase[j] = (unsigned)(pix < 0 ? 0 : pix) * (unsigned)p.C * 4u;
    }
    const int Ktot = p.KH * p.KW * p.C;
    unsigned bbase[BV];
#pragma unroll
    for (int j = 0; j < BV; ++j) {
        const int n = n0 + lrow + 32 * j;
        bbase[j] = n < p.N ? (unsigned)n * (unsigned)Ktot * 4u : p.wbytes;  // past the end -> zeros
    }

    // K-step state, advanced incrementally (channel block fastest, then kx, then ky over the class's taps)
    int st_cb, st_kx, st_ky;
    {
        const int tap = ks_begin / p.cblocks;
        st_cb = ks_begin - tap * p.cblocks;
        const int ty = tap / ntx;
        st_ky = pc.ky0 + ty;
        st_kx = pc.kx0 + (tap - ty * ntx);
    }
    float4 ra[AV], rb[BV];
    auto load_tile = [&]() {  // loads the tile of the current K-step state, then advances the state
        const int dy = (st_ky - (p.KH >> 1)) * p.dil, dx = (st_kx - (p.KW >> 1)) * p.dil;
        const int c = st_cb * BK + lcol;
        const bool cok = c < p.C;
        const int aoff = ((dy * p.W + dx) * p.C + c) * 4;
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            const int yy = ay[j] + dy, xx = ax[j] + dx;
            const bool ok = cok & (yy >= 0) & (yy < p.H) & (xx >= 0) & (xx < p.W);
            const unsigned off = ok ? abase[j] + (unsigned)aoff : p.xbytes;
            ra[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
        const unsigned boff = (unsigned)(((st_ky * p.KW + st_kx) * p.C + c) * 4);
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            const unsigned off = cok ? bbase[j] + boff : p.wbytes;
            rb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0));
        }
        if (++st_cb == p.cblocks) {
            st_cb = 0;
            if (++st_kx > pc.kx1) {
                st_kx = pc.kx0;
                ++st_ky;
            }
        }
    };
    // bf16x3: a row keeps its 144-byte stride: [32 hi bf16 | 32 lo bf16 | 16 B pad]
    auto store_split = [&](float *row, const float4 &v4) {
        const f32x4 v = {v4.x, v4.y, v4.z, v4.w};
        const bf16x4 hi = __builtin_convertvector(v, bf16x4);
        const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
        __bf16 *r = reinterpret_cast<__bf16 *>(row);
        *reinterpret_cast<bf16x4 *>(r + lcol) = hi;
        *reinterpret_cast<bf16x4 *>(r + BK + lcol) = lo;
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            if constexpr (MATH == MATH_BF16X3) store_split(&As[(lrow + 32 * j) * LDS_STRIDE], ra[j]);
            else *reinterpret_cast<float4 *>(&As[(lrow + 32 * j) * LDS_STRIDE + lcol]) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            if constexpr (MATH == MATH_BF16X3) store_split(&Bs[(lrow + 32 * j) * LDS_STRIDE], rb[j]);
            else *reinterpret_cast<float4 *>(&Bs[(lrow + 32 * j) * LDS_STRIDE + lcol]) = rb[j];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frag = (lane & 31) * LDS_STRIDE + (lane >> 5) * 4;
    const float *Aw = As + (wm * TM * 32) * LDS_STRIDE + frag;
    const float *Bw = Bs + (wn * TN * 32) * LDS_STRIDE + frag;

    auto compute_tile = [&]() {
        if constexpr (MATH == MATH_BF16X3) {
            // lane (i = l & 31, g = l >> 5) supplies k = 16*slice + 8*g .. +7 of row i for both operands (the same
            // k assignment on the A and the B side is all the dot product needs)
            const __bf16 *Ah = reinterpret_cast<const __bf16 *>(As + (wm * TM * 32 + (lane & 31)) * LDS_STRIDE) +
                               (lane >> 5) * 8;
            const __bf16 *Bh = reinterpret_cast<const __bf16 *>(Bs + (wn * TN * 32 + (lane & 31)) * LDS_STRIDE) +
                               (lane >> 5) * 8;
#pragma unroll
            for (int sl = 0; sl < BK / 16; ++sl) {
                bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = *reinterpret_cast<const bf16x8 *>(Ah + i * 32 * LDS_STRIDE * 2 + sl * 16);
                    al[i] = *reinterpret_cast<const bf16x8 *>(Ah + i * 32 * LDS_STRIDE * 2 + BK + sl * 16);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[j] = *reinterpret_cast<const bf16x8 *>(Bh + j * 32 * LDS_STRIDE * 2 + sl * 16);
                    bl[j] = *reinterpret_cast<const bf16x8 *>(Bh + j * 32 * LDS_STRIDE * 2 + BK + sl * 16);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
            return;
        }
#pragma unroll
        for (int kb = 0; kb < BK / 8; ++kb) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4 *>(Aw + i * 32 * LDS_STRIDE + kb * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4 *>(Bw + j * 32 * LDS_STRIDE + kb * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
    };

    // K loop; last tile peeled so the body has no conditional.  ks_begin >= ks_end only for an empty split-K slice
    // (block-uniform).
    if (ks_begin < ks_end) {
        load_tile();
        store_tile();
        __syncthreads();
        for (int ks = ks_begin; ks < ks_end - 1; ++ks) {
            load_tile();
            compute_tile();
            __syncthreads();
            store_tile();
            __syncthreads();
        }
        compute_tile();
    }

    // The 16-pass fp32 MFMA needs 18 wait states before its result is read.  hipcc (ROCm 7.2) was seen to place
    // the first v_accvgpr_read too early on a loop-exit edge (wrong last accumulator element); the wait is made
    // explicit here and tied to the accumulators so nothing is scheduled across it.
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[i][j]));

    // epilogue.  C/D layout of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).
    const int col = lane & 31, rsub = (lane >> 5) * 4;
    float *dst = p.splits == 1 ? p.y : p.ws + (size_t)si * p.M * p.N;
    const bool fused = p.splits == 1;
    int dy_, dx_;
    if ((p.N & 3) == 0) {
        // Vector path: each wave transposes one 32x32 accumulator tile at a time through its own 4.5 KiB LDS
        // slice so that a lane owns 4 consecutive channels of a pixel: 16-byte residual loads / output stores
        // (4x fewer memory instructions than the element-per-lane layout; the K = 256 conv3 layers are
        // epilogue-bound otherwise).
        __syncthreads();  // every wave is done with the last A/B tile
        float *ep = lds + wave * (32 * LDS_STRIDE);
        const int erow = lane >> 3, ecol = (lane & 7) * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int pix[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) pix[it] = decode(r0 + (wm * TM + i) * 32 + erow + 8 * it, dy_, dx_);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wn * TN + j) * 32 + ecol;
                float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
                if (fused && p.bias && n < p.N) bias = *reinterpret_cast<const float4 *>(p.bias + n);
#pragma unroll
                for (int e = 0; e < 16; ++e) ep[((e & 3) + 8 * (e >> 2) + rsub) * LDS_STRIDE + col] = acc[i][j][e];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    float4 v = *reinterpret_cast<const float4 *>(&ep[(erow + 8 * it) * LDS_STRIDE + ecol]);
                    if (pix[it] >= 0 && n < p.N) {
                        const size_t o = (size_t)pix[it] * p.N + n;
                        if (fused) {
                            v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
                            if (p.residual) {
                                const float4 rr = *reinterpret_cast<const float4 *>(p.residual + o);
                                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                            }
                            if (p.relu) {
                                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                            }
                        }
                        *reinterpret_cast<float4 *>(dst + o) = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        return;
    }
    // scalar path (N not a multiple of 4: the 3-channel xyz head, the 27- and 2-wide head outputs)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int pix = decode(r0 + (wm * TM + i) * 32 + rsub + (e & 3) + 8 * (e >> 2), dy_, dx_);
            if (pix < 0) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wn * TN + j) * 32 + col;
                if (n >= p.N) continue;
                const size_t o = (size_t)pix * p.N + n;
                float v = acc[i][j][e];
                if (fused) {
                    if (p.bias) v += p.bias[n];
                    if (p.residual) v += p.residual[o];
                    if (p.relu) v = fmaxf(v, 0.f);
                }
                dst[o] = v;
            }
        }
    }
}

// y = act(sum_s ws[s] + bias + residual)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ ws, int splits, long long MN,
                                                            int N, const float *__restrict__ bias,
                                                            const float *__restrict__ residual, int relu,
                                                            float *__restrict__ y)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < MN;
         i += (long long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < splits; ++s) v += ws[(size_t)s * MN + i];
        if (bias) v += bias[i % N];
        if (residual) v += residual[i];
        if (relu) v = fmaxf(v, 0.f);
        y[i] = v;
    }
}

// Pixel classes of a layer.  Plain layers get one class covering the image with every tap (border handled by the
// per-row bounds tests).  Atrous 3x3 layers with room for it get up to 3 x 3 classes, longest K loops first so the
// dispatcher starts the heavy tiles early.
void build_classes(ConvParams &p, int B, int BM, bool use_classes)
{
    struct Span {
        int o, len, k0, k1;
    };
    Span ys[3], xs[3];
    int ny = 0, nx = 0;
    const int d = p.dil, hk = p.KH >> 1, wk = p.KW >> 1;
    if (use_classes && p.KH == 3 && p.H >= 2 * d) {
        if (p.H - 2 * d > 0) ys[ny++] = {d, p.H - 2 * d, 0, 2};
        ys[ny++] = {0, d, hk, 2};
        ys[ny++] = {p.H - d, d, 0, hk};
    } else {
        ys[ny++] = {0, p.H, 0, p.KH - 1};
    }
    if (use_classes && p.KW == 3 && p.W >= 2 * d) {
        if (p.W - 2 * d > 0) xs[nx++] = {d, p.W - 2 * d, 0, 2};
        xs[nx++] = {0, d, wk, 2};
        xs[nx++] = {p.W - d, d, 0, wk};
    } else {
        xs[nx++] = {0, p.W, 0, p.KW - 1};
    }
    // enumerate, then order by tap count (most first); at most 9 entries, so a selection sort will do
    p.ncls = 0;
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) {
            PixelClass &c = p.cls[p.ncls++];
            c.y0 = ys[a].o; c.h = ys[a].len; c.ky0 = ys[a].k0; c.ky1 = ys[a].k1;
            c.x0 = xs[b].o; c.w = xs[b].len; c.kx0 = xs[b].k0; c.kx1 = xs[b].k1;
            c.rows = B * c.h * c.w;
            c.tiles = mpsr::ceil_div(c.rows, BM);
        }
    auto taps = [](const PixelClass &c) { return (c.ky1 - c.ky0 + 1) * (c.kx1 - c.kx0 + 1); };
    for (int i = 0; i < p.ncls; ++i)
        for (int j = i + 1; j < p.ncls; ++j)
            if (taps(p.cls[j]) > taps(p.cls[i])) {
                const PixelClass tmp = p.cls[i];
                p.cls[i] = p.cls[j];
                p.cls[j] = tmp;
            }
    p.mtiles_xcd = 0;
    for (int i = 0; i < p.ncls; ++i) p.mtiles_xcd += mpsr::ceil_div(p.cls[i].tiles, 8);
}

template <int BM, int BN, int WM, int WN, int MATH = MATH_FP32>
int launch(ConvParams &p, int B, bool use_classes, hipStream_t s)
{
    build_classes(p, B, BM, use_classes);
    p.ntiles = mpsr::ceil_div(p.N, BN);
    const long long blocks = 8LL * p.mtiles_xcd * p.ntiles * p.splits;
    if (blocks > 0x7fffffffLL) return mpsr::fail(MPSR_ERR_UNSUPPORTED, "conv2d: grid too large");
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, MATH>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    MPSR_CHECK_LAUNCH("conv_igemm_kernel");
    return MPSR_OK;
}

// tuning knobs (tests sweep them): tile -1 = heuristic, 0=128x128 1=128x64 2=64x128 3=64x64 4=128x32 5=96x128;
// classes -1 = heuristic, 0 = never, 1 = whenever the geometry allows
int g_tile_override = -1;
int g_class_override = -1;
int g_math = MATH_FP32;  // mpsr_set_conv_math

}  // namespace

namespace mpsr {

int conv3x3_narrow(const float *x, int B, int H, int W, int C, const float *w, const float *bias, int relu, float *y,
                   int N, hipStream_t s);  // image_ops.hip

// split_k == 0 ("auto"): a launch whose 64x64 tiles would leave most of the 256 CUs with one workgroup or none
// (32 proposal crops: M = 4608; the 40x152 full-image map: M = 6080) is cut along K so that ~3 workgroups per CU
// exist, each keeping at least 4 K steps; bounded by the scratch the caller provides.  kAutoSplitFloats covers every
// case the rule can produce: (768 + tiles) * 64 * 64 floats with tiles < 768.
// Measured at 32 crops (tools/conv_layer_bench.py --batch 32 --split 0): 768 workgroups / >= 4 K steps per slice is
// the sweet spot (conv time 5.02 -> 4.47 ms); asking for 1536+ workgroups loses again to partial-sum traffic.
size_t conv_auto_split_floats() { return (size_t)1536 * 64 * 64; }

constexpr int g_auto_split_target = 768;  // workgroups wanted per launch
constexpr int g_auto_split_min_ksteps = 4;

int auto_split_k(int M, int N, int ksteps, const float *ws, size_t ws_floats)
{
    if (!ws) return 1;
    const long long tiles = (long long)ceil_div(M, 64) * ceil_div(N, 64);
    if (tiles >= g_auto_split_target || N <= 32) return 1;
    long long s = (g_auto_split_target + tiles - 1) / tiles;
    if (s > 8) s = 8;
    if (s > ksteps / g_auto_split_min_ksteps) s = ksteps / g_auto_split_min_ksteps;
    while (s > 1 && (size_t)s * M * N > ws_floats) --s;
    return s < 1 ? 1 : (int)s;
}

// Shared by the network-level entry points (network.hip).
int conv2d(const float *x, int B, int H, int W, int C, const float *w, const float *bias, const float *residual,
           float *y, int N, int KH, int KW, int dilation, int relu, int split_k, float *ws, size_t ws_floats,
           hipStream_t stream)
{
    MPSR_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && N > 0, "conv2d: bad shape (B=%d H=%d W=%d C=%d N=%d)", B, H, W,
                 C, N);
    MPSR_REQUIRE(KH >= 1 && KW >= 1 && (KH & 1) && (KW & 1) && dilation >= 1,
                 "conv2d: kernel %dx%d (odd sizes only) dilation %d", KH, KW, dilation);
    MPSR_REQUIRE(C % 4 == 0, "conv2d: C=%d must be a multiple of 4 (pad the channels)", C);
    if (B == 0) return MPSR_OK;
    MPSR_REQUIRE(x && w && y, "conv2d: null pointer");
    MPSR_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "conv2d: x and w must be 16-byte aligned");
    const long long M64 = (long long)B * H * W;
    // 32-bit buffer offsets: activations and weights must each stay below 4 GiB (B=256 at 48x48x256 is 0.6 GiB)
    MPSR_REQUIRE(M64 * C * 4 < 0xfffffff0LL && (long long)N * KH * KW * C * 4 < 0xfffffff0LL && M64 < 0x7fffffffLL,
                 "conv2d: tensor exceeds the 4 GiB addressable by one launch; split the batch");
    // N <= 4 3x3 layers (the xyz-map head) are input-bandwidth-bound: direct VALU kernel instead of a 32-wide MFMA
    // tile (an explicit tile override keeps them on the MFMA path so tests cover both)
    if (KH == 3 && KW == 3 && dilation == 1 && N <= 4 && C % 32 == 0 && !residual && split_k <= 1 &&
        g_tile_override < 0 && ((uintptr_t)w & 3) == 0)
        return conv3x3_narrow(x, B, H, W, C, w, bias, relu, y, N, stream);
    ConvParams p;
    p.x = x; p.w = w; p.bias = bias; p.residual = residual; p.y = y; p.ws = ws;
    p.M = (int)M64; p.H = H; p.W = W; p.C = C; p.N = N; p.KH = KH; p.KW = KW; p.dil = dilation; p.relu = relu;
    p.xbytes = (unsigned)(M64 * C * 4);
    p.wbytes = (unsigned)((long long)N * KH * KW * C * 4);
    p.cblocks = ceil_div(C, BK);
    p.ksteps_total = KH * KW * p.cblocks;
    if (split_k == 0) split_k = auto_split_k(p.M, N, p.ksteps_total, ws, ws_floats);  // fill an under-filled launch
    if (split_k < 1) split_k = 1;
    if (split_k > p.ksteps_total) split_k = p.ksteps_total;
    p.ksteps_per_split = ceil_div(p.ksteps_total, split_k);
    p.splits = ceil_div(p.ksteps_total, p.ksteps_per_split);
    if (p.splits > 1) {
        if (!ws || ws_floats < (size_t)p.splits * p.M * N)
            return fail(MPSR_ERR_WORKSPACE, "conv2d: split_k=%d needs %zu workspace floats, got %zu", p.splits,
                        (size_t)p.splits * p.M * N, ws_floats);
    }
    // border-class tiling: atrous 3x3 layers only (a 1-pixel border ring at dilation 1 saves too little)
    bool use_classes = KH == 3 && KW == 3 && dilation > 1;
    if (g_class_override == 0) use_classes = false;
    if (g_class_override == 1) use_classes = KH == 3 && KW == 3;
    int rc;
    int sel = g_tile_override;
    if (sel < 0) {
        // measured on MI355X (tools/conv_layer_bench.py, profiles/): 128x128 wins once there are >= ~18 tiles per CU
        // (the 24x24 / 48x48 decoder layers, 130-137 TFLOP/s); the 12x12 trunk layers (M = 36864 at B = 256) only
        // make 2.25 tiles of 128x128 per CU, so 64x64 tiles (9 per CU, 7 waves/SIMD) balance and overlap better
        if (N <= 32) sel = 4;
        else if (p.M < 131072) sel = 3;
        else sel = 0;
    }
    if (g_math == MATH_BF16X3) {
        // 3 bf16 matrix instructions replace 8 fp32 ones at 1/16 of the cycles each, so LDS and the operand split
        // feed the matrix pipes: larger wave tiles (fewer fragment reads per instruction) win -- 128x128 on the big
        // decoder maps, 128x64 on the 12x12 trunk layers (tools/conv_layer_bench.py --math bf16x3)
        // (small batches -- a single image's 32 crops -- need the small tile to have workgroups for every CU)
        if (g_tile_override < 0) sel = N <= 32 ? 4 : (p.M < 16384 ? 3 : ((N <= 64 || p.M < 131072) ? 1 : 0));
        switch (sel) {
            case 0: rc = launch<128, 128, 2, 2, MATH_BF16X3>(p, B, use_classes, stream); break;
            case 1: rc = launch<128, 64, 2, 2, MATH_BF16X3>(p, B, use_classes, stream); break;
            case 2: rc = launch<64, 128, 2, 2, MATH_BF16X3>(p, B, use_classes, stream); break;
            case 3: rc = launch<64, 64, 2, 2, MATH_BF16X3>(p, B, use_classes, stream); break;
            case 5: rc = launch<96, 128, 1, 4, MATH_BF16X3>(p, B, use_classes, stream); break;
            default: rc = launch<128, 32, 4, 1, MATH_BF16X3>(p, B, use_classes, stream); break;
        }
    } else {
        switch (sel) {
            case 0: rc = launch<128, 128, 2, 2>(p, B, use_classes, stream); break;
            case 1: rc = launch<128, 64, 2, 2>(p, B, use_classes, stream); break;
            case 2: rc = launch<64, 128, 2, 2>(p, B, use_classes, stream); break;
            case 3: rc = launch<64, 64, 2, 2>(p, B, use_classes, stream); break;
            case 5: rc = launch<96, 128, 1, 4>(p, B, use_classes, stream); break;
            default: rc = launch<128, 32, 4, 1>(p, B, use_classes, stream); break;
        }
    }
    if (rc) return rc;
    if (p.splits > 1) {
        const long long MN = (long long)p.M * N;
        const int grid = (int)((MN + 255) / 256 < 65536 ? (MN + 255) / 256 : 65536);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, stream, ws, p.splits, MN, N, bias, residual,
                           relu, y);
        MPSR_CHECK_LAUNCH("splitk_reduce_kernel");
    }
    return MPSR_OK;
}

}  // namespace mpsr

// Internal (not part of the ABI in monopsr_hip.h): force a tile configuration / the border-class tiling for tuning
// and for the tests that sweep every instantiation.  Process-wide, not thread-safe.
extern "C" void mpsr_debug_set_conv_tile(int sel) { g_tile_override = sel; }
extern "C" void mpsr_debug_set_conv_classes(int mode) { g_class_override = mode; }

extern "C" int mpsr_set_conv_math(int mode)
{
    MPSR_REQUIRE(mode == MATH_FP32 || mode == MATH_BF16X3, "set_conv_math: unknown mode %d", mode);
    g_math = mode;
    return MPSR_OK;
}
extern "C" int mpsr_get_conv_math(void) { return g_math; }

extern "C" int mpsr_conv2d_nhwc_f32(const float *x, int B, int H, int W, int C, const float *w, const float *bias,
                                    const float *residual, float *y, int N, int KH, int KW, int dilation, int relu,
                                    int split_k, float *ws, size_t ws_floats, mpsr_stream_t stream)
{
    return mpsr::conv2d(x, B, H, W, C, w, bias, residual, y, N, KH, KW, dilation, relu, split_k, ws, ws_floats,
                        mpsr::as_stream(stream));
}
