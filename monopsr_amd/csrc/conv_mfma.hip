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
#include <atomic>
#include <mutex>
#include <type_traits>

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

using mpsr::FastDiv;
using mpsr::fdiv;
using mpsr::make_fastdiv;

// A rectangle of output pixels [y0, y0+h) x [x0, x0+w) of every image whose in-image taps are ky0..ky1 x kx0..kx1.
struct PixelClass {
    int y0, h, x0, w;
    int ky0, ky1, kx0, kx1;
    int tiles;  // M tiles of this class
    int rows;   // B * h * w
    FastDiv fd_ppi, fd_w;  // by h * w and by w
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
    FastDiv fd_ntiles, fd_mtiles;
    unsigned xbytes, wbytes;
    size_t ws_floats;
    int ncls;
    PixelClass cls[MAX_CLS];
};

// PLAIN: a 1x1 layer (convolution or fully connected) over whole images with C a multiple of 32 -- row r of the GEMM
// is pixel r, a K step is the next 128 bytes of every row: no decode, and the K loop's loads take their per-step
// offset in a scalar register (no vector-ALU work at all between the MFMAs).
// KIND 2 (CLS): a 3x3 layer tiled by border classes with C a multiple of 32 -- every tap a class visits is inside the
// image for every row of the class, so the per-load bounds tests go too: a row's offset is fixed, the tap / channel
// block offset of a K step is one scalar (the descriptor's base is moved back by the most negative tap offset so that
// the scalar stays non-negative; the hardware range-checks the vector offset only).
template <int BM, int BN, int WM, int WN, int MATH = MATH_FP32, int DEPTH = 1, int KIND = 0>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p)
{
    constexpr bool PLAIN = KIND == 1, CLS = KIND == 2;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int AV = BM / 32, BV = BN / 32;  // float4 loads per thread per tile
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDS_STRIDE];
    float *As = lds, *Bs = lds + BM * LDS_STRIDE;
#ifdef MPSR_TRACE
    unsigned long long tr0 = __builtin_readcyclecounter(), tr1 = 0, tr2 = 0;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif

    // Tile order.  Workgroup b is observed to run on XCD b % 8 (speed only, never correctness): XCD x takes M tiles
    // x, x+8, x+16, ... of EVERY pixel class -- so each XCD gets the same mix of long and short K loops -- and walks
    // them heaviest class first, the N tiles of one M tile back to back so its activation panel stays in that
    // XCD's L2.  Workgroups past an XCD's share (classes whose tile count is not a multiple of 8) exit at once.
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int lq = fdiv(l, p.fd_ntiles);
    const int ni = l - lq * p.ntiles;
    const int si = fdiv(lq, p.fd_mtiles);
    int lm = lq - si * p.mtiles_xcd;
    const int n0 = ni * BN;
    int ci = -1;
    if constexpr (PLAIN) {
        if (xcd + 8 * lm < p.cls[0].tiles) ci = 0;
    } else {
        for (int c = 0; c < p.ncls; ++c) {
            const int cnt = p.cls[c].tiles > xcd ? (p.cls[c].tiles - xcd + 7) >> 3 : 0;
            if (lm < cnt) {
                ci = c;
                break;
            }
            lm -= cnt;
        }
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
        if constexpr (PLAIN) {
            yy = xx = 0;
            return row < pc.rows ? row : -1;
        }
        if (row >= pc.rows) {
            yy = -(1 << 20);  // fails every bounds test
            xx = 0;
            return -1;
        }
        const int img = fdiv(row, pc.fd_ppi), pp = row - img * ppi;
        const int py = fdiv(pp, pc.fd_w);
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
        // the whole voffset; rows past the end get one that is out of range for the (CLS: enlarged) descriptor
        if constexpr (PLAIN || CLS)
            abase[j] = pix < 0 ? p.xbytes + (CLS ? (unsigned)(p.dil * (p.W + 1) * p.C * 4) : 0u)
                               : abase[j] + (unsigned)lcol * 4u;
    }
    const int Ktot = p.KH * p.KW * p.C;
    unsigned bbase[BV];
#pragma unroll
    for (int j = 0; j < BV; ++j) {
        const int n = n0 + lrow + 32 * j;
        bbase[j] = n < p.N ? (unsigned)n * (unsigned)Ktot * 4u : p.wbytes;  // past the end -> zeros
        if constexpr (PLAIN || CLS) bbase[j] = n < p.N ? bbase[j] + (unsigned)lcol * 4u : p.wbytes;
    }

    // K-step state, advanced incrementally: taps fastest (kx, then ky over the class's taps), channel block slowest.
    // The taps of one 32-channel block re-read the same few KB of pixels (shifted), so their gathers hit L1 / L2;
    // with the channel block fastest a pixel came back after a whole channel sweep (1/9 of the K loop, far more than
    // the XCD's L2 holds across its resident tiles) and the 3x3 layers fetched their input ~4.7x from HBM.
    int st_cb, st_kx, st_ky;
    if (PLAIN || ks_begin == 0) {  // block-uniform; PLAIN has one tap, so st_cb counts K steps
        st_cb = ks_begin;
        st_ky = pc.ky0;
        st_kx = pc.kx0;
    } else {
        const int ntaps = (pc.ky1 - pc.ky0 + 1) * ntx;
        st_cb = ks_begin / ntaps;
        const int tap = ks_begin - st_cb * ntaps;
        const int ty = tap / ntx;
        st_ky = pc.ky0 + ty;
        st_kx = pc.kx0 + (tap - ty * ntx);
    }
    // register staging: one tile in flight (DEPTH 1) or two (DEPTH 2: the loads of K step k + 2 are issued while step k
    // computes, so a load has a whole K step of every resident wave to return -- what a launch's last tiles need,
    // when only 2-4 waves per SIMD are left to hide the ~1.5 us the memory system takes under this load)
    float4 ra[DEPTH][AV], rb[DEPTH][BV];
    // loads the tile of the current K-step state into staging set `set` (a literal at every call), then advances
    // the state; `live` false (past the last K step) turns every load into an out-of-range one (no traffic)
    auto load_tile = [&](auto set_c, bool live) {
        constexpr int set = decltype(set_c)::value;
        if constexpr (PLAIN) {
            const int soff = st_cb * (BK * 4);
            // a dead load (DEPTH 2 runs past the last K step) goes through a zero-length descriptor: a scalar
            // select instead of a per-lane one
            const __amdgpu_buffer_rsrc_t rxl =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, live ? (int)p.xbytes : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rwl =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, live ? (int)p.wbytes : 0, 0x00020000);
#pragma unroll
            for (int j = 0; j < AV; ++j)
                ra[set][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rxl, abase[j], soff, 0));
#pragma unroll
            for (int j = 0; j < BV; ++j)
                rb[set][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rwl, bbase[j], soff, 0));
            ++st_cb;
            return;
        }
        if constexpr (CLS) {
            const int dy = (st_ky - (p.KH >> 1)) * p.dil, dx = (st_kx - (p.KW >> 1)) * p.dil;
            const int bias = p.dil * (p.W + 1) * p.C * 4;  // -(most negative tap offset)
            const int soffa = ((dy * p.W + dx) * p.C + st_cb * BK) * 4 + bias;
            const int soffb = ((st_ky * p.KW + st_kx) * p.C + st_cb * BK) * 4;
            // the window starts `bias` bytes before the tensor, so it is `bias` bytes longer: a valid row + tap lands
            // in [bias, bias + xbytes) whether the hardware range-checks the scalar offset or not
            const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char *>(reinterpret_cast<const char *>(p.x)) - bias, 0, live ? (int)p.xbytes + bias : 0,
                0x00020000);
            const __amdgpu_buffer_rsrc_t rwl =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, live ? (int)p.wbytes : 0, 0x00020000);
#pragma unroll
            for (int j = 0; j < AV; ++j)
                ra[set][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rxl, abase[j], soffa, 0));
#pragma unroll
            for (int j = 0; j < BV; ++j)
                rb[set][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rwl, bbase[j], soffb, 0));
            if (++st_kx > pc.kx1) {
                st_kx = pc.kx0;
                if (++st_ky > pc.ky1) {
                    st_ky = pc.ky0;
                    ++st_cb;
                }
            }
            return;
        }
        const int dy = (st_ky - (p.KH >> 1)) * p.dil, dx = (st_kx - (p.KW >> 1)) * p.dil;
        const int c = st_cb * BK + lcol;
        const bool cok = (c < p.C) & live;
        const int aoff = ((dy * p.W + dx) * p.C + c) * 4;
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            const int yy = ay[j] + dy, xx = ax[j] + dx;
            const bool ok = cok & (yy >= 0) & (yy < p.H) & (xx >= 0) & (xx < p.W);
            const unsigned off = ok ? abase[j] + (unsigned)aoff : p.xbytes;
            ra[set][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
        const unsigned boff = (unsigned)(((st_ky * p.KW + st_kx) * p.C + c) * 4);
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            const unsigned off = cok ? bbase[j] + boff : p.wbytes;
            rb[set][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0));
        }
        if (++st_kx > pc.kx1) {
            st_kx = pc.kx0;
            if (++st_ky > pc.ky1) {
                st_ky = pc.ky0;
                ++st_cb;
            }
        }
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, DEPTH - 1>;
    // bf16x3: a row keeps its 144-byte stride: [32 hi bf16 | 32 lo bf16 | 16 B pad]
    auto store_split = [&](float *row, const float4 &v4) {
        const f32x4 v = {v4.x, v4.y, v4.z, v4.w};
        const bf16x4 hi = __builtin_convertvector(v, bf16x4);
        const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
        __bf16 *r = reinterpret_cast<__bf16 *>(row);
        *reinterpret_cast<bf16x4 *>(r + lcol) = hi;
        *reinterpret_cast<bf16x4 *>(r + BK + lcol) = lo;
    };
    auto store_tile = [&](auto set_c) {
        constexpr int set = decltype(set_c)::value;
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            if constexpr (MATH == MATH_BF16X3) store_split(&As[(lrow + 32 * j) * LDS_STRIDE], ra[set][j]);
            else *reinterpret_cast<float4 *>(&As[(lrow + 32 * j) * LDS_STRIDE + lcol]) = ra[set][j];
        }
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            if constexpr (MATH == MATH_BF16X3) store_split(&Bs[(lrow + 32 * j) * LDS_STRIDE], rb[set][j]);
            else *reinterpret_cast<float4 *>(&Bs[(lrow + 32 * j) * LDS_STRIDE + lcol]) = rb[set][j];
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
        load_tile(Set0{}, true);
        store_tile(Set0{});
        __syncthreads();
#ifdef MPSR_TRACE
        tr1 = __builtin_readcyclecounter();
#endif
        if constexpr (DEPTH == 1) {
            for (int ks = ks_begin; ks < ks_end - 1; ++ks) {
                load_tile(Set0{}, true);
                compute_tile();
                __syncthreads();
                store_tile(Set0{});
                __syncthreads();
            }
            compute_tile();
        } else {
            // LDS holds step k; set 0 holds step k + 1 (in flight); two steps per trip, each staging set a literal
            const int n = ks_end - ks_begin;
            load_tile(Set0{}, n > 1);
            int k = 0;
            for (; k + 2 < n; k += 2) {
                load_tile(Set1{}, true);  // step k + 2
                compute_tile();
                __syncthreads();
                store_tile(Set0{});  // step k + 1
                __syncthreads();
                load_tile(Set0{}, k + 3 < n);  // step k + 3
                compute_tile();
                __syncthreads();
                store_tile(Set1{});  // step k + 2
                __syncthreads();
            }
            compute_tile();
            if (n - k == 2) {  // block-uniform
                __syncthreads();
                store_tile(Set0{});
                __syncthreads();
                compute_tile();
            }
        }
    }

    // The 16-pass fp32 MFMA needs 18 wait states before its result is read.  hipcc (ROCm 7.2) was seen to place
    // the first v_accvgpr_read too early on a loop-exit edge (wrong last accumulator element); the wait is made
    // explicit here and tied to the accumulators so nothing is scheduled across it.
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[i][j]));

#ifdef MPSR_TRACE
    tr2 = __builtin_readcyclecounter();
    auto trace_end = [&]() {
        if (threadIdx.x == 0 && p.ws) {
            unsigned long long *r = reinterpret_cast<unsigned long long *>(p.ws) + (size_t)blockIdx.x * 8;
            r[0] = tr0; r[1] = tr1; r[2] = tr2; r[3] = __builtin_readcyclecounter();
            r[4] = __builtin_amdgcn_s_getreg(63492);  // HW_REG_HW_ID
            r[5] = __builtin_amdgcn_s_getreg(63508);  // HW_REG_XCC_ID
            r[6] = rt0; r[7] = __builtin_amdgcn_s_memrealtime();
        }
    };
#endif
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
#ifdef MPSR_TRACE
        trace_end();
#endif
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

// ------------------------------------------------------------------------------------------------ stream-K
// Persistent form of the same implicit GEMM.  A launch's work is the sequence of (tile, K step) "units" in tile order
// (pixel classes heaviest first, the N tiles of an M tile back to back); exactly G = CUs x resident-workgroups-per-CU
// workgroups are launched -- all resident at once -- and workgroup v takes the contiguous unit range
// [bound(v), bound(v+1)), i.e. every workgroup gets the same number of K steps whatever the tile count, the class mix
// or the tail.  (One tile per workgroup leaves the 12x12 layers' 2304 tiles on 1792 resident slots: the last 512 tiles
// start when the first slots free up and the chip idles behind them.)
// A range that ends inside a tile leaves a PARTIAL accumulator: it is written to the workgroup's slab, the
// workgroup takes a ticket on the tile's arrival counter, and whichever contributor arrives LAST adds the partials in
// K order (own registers in their place, so the sum does not depend on who was last) and runs the fused epilogue.
// Nobody ever waits for another workgroup, so the kernel cannot deadlock whatever the residency turns out to be.
// Inter-workgroup visibility follows cdna_hip_programming.md Guideline 16: plain slab stores, every wave drains
// vmcnt, barrier, one lane's agent-scope release (+ an asm vmcnt(0) the compiler cannot drop) before the relaxed
// agent-scope ticket; the last arriver's one lane does an agent-scope acquire, barrier, then plain loads.
struct SkParams {
    int G;                          // persistent workgroups, a multiple of 8
    unsigned U;                     // units in the launch
    unsigned q, r;                  // U = G * q + r: bound(v) = v * q + min(v, r)
    float *slabs;                   // 2 slabs of BM * BN floats per workgroup (a partial that starts the workgroup's
                                    // range: slot 0; one that ends it: slot 1)
    int *counters;                  // one arrival counter per tile; zero on entry, zero again on exit
    unsigned ubase[MAX_CLS + 1];    // first unit of class c
    unsigned tbase[MAX_CLS + 1];    // first tile id of class c
    int ks[MAX_CLS];                // K steps per tile of class c
};

__device__ __forceinline__ unsigned sk_bound(const SkParams &s, unsigned v) { return v * s.q + min(v, s.r); }

// the workgroup whose range holds unit u (u < U)
__device__ __forceinline__ unsigned sk_owner(const SkParams &s, unsigned u)
{
    const unsigned thr = s.r * (s.q + 1);
    return u < thr ? u / (s.q + 1) : s.r + (u - thr) / s.q;
}

template <int BM, int BN, int WM, int WN, int MATH = MATH_FP32>
__global__ __launch_bounds__(256) void conv_sk_kernel(const ConvParams p, const SkParams sk)
{
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int AV = BM / 32, BV = BN / 32;
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDS_STRIDE + 4];
    float *As = lds, *Bs = lds + BM * LDS_STRIDE;
    int *flag = reinterpret_cast<int *>(lds + (BM + BN) * LDS_STRIDE);

    // workgroup b is observed on XCD b % 8 (speed only): give each XCD one contiguous eighth of the unit sequence, so
    // the workgroups sharing an L2 work on neighbouring M panels
    const unsigned v = (blockIdx.x & 7) * ((unsigned)sk.G >> 3) + (blockIdx.x >> 3);
    const unsigned ustart = sk_bound(sk, v), uend = sk_bound(sk, v + 1);
    unsigned u = ustart;

    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, (int)p.wbytes, 0x00020000);
    const int Ktot = p.KH * p.KW * p.C;

    while (u < uend) {
        // the thread id is made opaque per segment so that nothing derived from it is hoisted out of this loop and
        // kept alive across it (that costs 14-20 VGPRs, i.e. one to two waves per SIMD of occupancy)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6;
        const int wm = wave / WN, wn = wave % WN;
        const int lrow = tid >> 3, lcol = (tid & 7) * 4;
        const int frag = (lane & 31) * LDS_STRIDE + (lane >> 5) * 4;
        const float *Aw = As + (wm * TM * 32) * LDS_STRIDE + frag;
        const float *Bw = Bs + (wn * TN * 32) * LDS_STRIDE + frag;
        int ci = 0;
        while (u >= sk.ubase[ci + 1]) ++ci;
        const PixelClass pc = p.cls[ci];
        const int ksc = sk.ks[ci];
        const unsigned local = u - sk.ubase[ci];
        const unsigned tic = local / (unsigned)ksc;  // tile inside its class
        const int ks_begin = (int)(local - tic * (unsigned)ksc);
        const int ks_end = min(ksc, ks_begin + (int)(uend - u));
        const int mt = (int)(tic / (unsigned)p.ntiles), ni = (int)(tic - (unsigned)mt * (unsigned)p.ntiles);
        const int r0 = mt * BM, n0 = ni * BN;
        const int ppi = pc.h * pc.w;
        const int ntx = pc.kx1 - pc.kx0 + 1;

        auto decode = [&](int row, int &yy, int &xx) -> int {
            if (row >= pc.rows) {
                yy = -(1 << 20);
                xx = 0;
                return -1;
            }
            const int img = fdiv(row, pc.fd_ppi), pp = row - img * ppi;
            const int py = fdiv(pp, pc.fd_w);
            yy = pc.y0 + py;
            xx = pc.x0 + (pp - py * pc.w);
            return (img * p.H + yy) * p.W + xx;
        };
        int ay[AV], ax[AV];
        unsigned abase[AV];
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            const int pix = decode(r0 + lrow + 32 * j, ay[j], ax[j]);
            abase[j] = (unsigned)(pix < 0 ? 0 : pix) * (unsigned)p.C * 4u;
        }
        unsigned bbase[BV];
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            const int n = n0 + lrow + 32 * j;
            bbase[j] = n < p.N ? (unsigned)n * (unsigned)Ktot * 4u : p.wbytes;
        }
        int st_cb, st_kx, st_ky;
        {
            const int ntaps = (pc.ky1 - pc.ky0 + 1) * ntx;
            st_cb = ks_begin / ntaps;
            const int tap = ks_begin - st_cb * ntaps;
            const int ty = tap / ntx;
            st_ky = pc.ky0 + ty;
            st_kx = pc.kx0 + (tap - ty * ntx);
        }
        float4 ra[AV], rb[BV];
        auto load_tile = [&]() {
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
            if (++st_kx > pc.kx1) {
                st_kx = pc.kx0;
                if (++st_ky > pc.ky1) {
                    st_ky = pc.ky0;
                    ++st_cb;
                }
            }
        };
        auto store_split = [&](float *row, const float4 &v4) {
            const f32x4 vv = {v4.x, v4.y, v4.z, v4.w};
            const bf16x4 hi = __builtin_convertvector(vv, bf16x4);
            const bf16x4 lo = __builtin_convertvector(vv - __builtin_convertvector(hi, f32x4), bf16x4);
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

        auto compute_tile = [&]() {
            if constexpr (MATH == MATH_BF16X3) {
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

        // K loop over this segment (always at least one step)
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
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[i][j]));

        const unsigned seg_first = u;
        u += (unsigned)(ks_end - ks_begin);
        bool finish = true;  // this workgroup runs the tile's epilogue
        // partial tile: publish the raw accumulators, take a ticket; the last arriver reduces in the epilogue
        int pieces = 1;
        unsigned t0 = 0, vfirst = 0;
        if (ks_begin != 0 || ks_end != ksc) {
            t0 = sk.ubase[ci] + tic * (unsigned)ksc;
            vfirst = sk_owner(sk, t0);
            pieces = (int)(sk_owner(sk, t0 + (unsigned)ksc - 1) - vfirst) + 1;
            int *cnt = sk.counters + (sk.tbase[ci] + tic);
            float *slab = sk.slabs + (size_t)(2 * v + (seg_first == ustart ? 0 : 1)) * (BM * BN);
            // lane-major float4: [wave][i][j][e / 4][lane]
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const float4 val = make_float4(acc[i][j][4 * e4], acc[i][j][4 * e4 + 1], acc[i][j][4 * e4 + 2],
                                                       acc[i][j][4 * e4 + 3]);
                        *reinterpret_cast<float4 *>(slab + ((((wave * TM + i) * TN + j) * 4 + e4) * 64 + lane) * 4) = val;
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                *flag = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            finish = *flag == pieces - 1;
            if (finish) {
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // clean for the next launch
                }
                __syncthreads();
            }
        }
        // accumulator tile (i, j) of this wave: its own registers for a whole tile; for a split tile the K-ordered
        // sum of every contributor's slab, this workgroup's own included (stored above) -- the same sum whoever
        // arrives last, and only one 32x32 tile of it is live at a time
        auto tile_values = [&](int i, int j, const f32x16 &own) -> f32x16 {
            if (pieces == 1) return own;
            f32x16 t;
            for (int qq = 0; qq < pieces; ++qq) {
                const unsigned vq = vfirst + (unsigned)qq;
                const float *src = sk.slabs + (size_t)(2 * vq + (sk_bound(sk, vq) < t0 ? 1 : 0)) * (BM * BN) +
                                   (((wave * TM + i) * TN + j) * 4 * 64 + lane) * 4;
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const float4 val = *reinterpret_cast<const float4 *>(src + e4 * 256);
                    if (qq == 0) {
                        t[4 * e4] = val.x; t[4 * e4 + 1] = val.y; t[4 * e4 + 2] = val.z; t[4 * e4 + 3] = val.w;
                    } else {
                        t[4 * e4] += val.x; t[4 * e4 + 1] += val.y; t[4 * e4 + 2] += val.z; t[4 * e4 + 3] += val.w;
                    }
                }
            }
            return t;
        };

        if (finish) {
            const int col = lane & 31, rsub = (lane >> 5) * 4;
            int dy_, dx_;
            if ((p.N & 3) == 0) {
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
                        if (p.bias && n < p.N) bias = *reinterpret_cast<const float4 *>(p.bias + n);
                        const f32x16 tv = tile_values(i, j, acc[i][j]);
#pragma unroll
                        for (int e = 0; e < 16; ++e) ep[((e & 3) + 8 * (e >> 2) + rsub) * LDS_STRIDE + col] = tv[e];
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            float4 vv = *reinterpret_cast<const float4 *>(&ep[(erow + 8 * it) * LDS_STRIDE + ecol]);
                            if (pix[it] >= 0 && n < p.N) {
                                const size_t o = (size_t)pix[it] * p.N + n;
                                vv.x += bias.x; vv.y += bias.y; vv.z += bias.z; vv.w += bias.w;
                                if (p.residual) {
                                    const float4 rr = *reinterpret_cast<const float4 *>(p.residual + o);
                                    vv.x += rr.x; vv.y += rr.y; vv.z += rr.z; vv.w += rr.w;
                                }
                                if (p.relu) {
                                    vv.x = fmaxf(vv.x, 0.f); vv.y = fmaxf(vv.y, 0.f);
                                    vv.z = fmaxf(vv.z, 0.f); vv.w = fmaxf(vv.w, 0.f);
                                }
                                *reinterpret_cast<float4 *>(p.y + o) = vv;
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int n = n0 + (wn * TN + j) * 32 + col;
                        const f32x16 tv = tile_values(i, j, acc[i][j]);
                        if (n >= p.N) continue;
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int pix = decode(r0 + (wm * TM + i) * 32 + rsub + (e & 3) + 8 * (e >> 2), dy_, dx_);
                            if (pix < 0) continue;
                            const size_t o = (size_t)pix * p.N + n;
                            float vv = tv[e];
                            if (p.bias) vv += p.bias[n];
                            if (p.residual) vv += p.residual[o];
                            if (p.relu) vv = fmaxf(vv, 0.f);
                            p.y[o] = vv;
                        }
                    }
                }
            }
        }
        __syncthreads();  // LDS (tiles, epilogue slices, flag) is reused by the next segment
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
            c.fd_ppi = make_fastdiv(c.h * c.w);
            c.fd_w = make_fastdiv(c.w);
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

std::atomic<int> g_plain_override{-1};  // 0: never use the PLAIN instantiation (tests / A-B), else whenever it applies

template <int BM, int BN, int WM, int WN, int MATH = MATH_FP32, int DEPTH = 1>
int launch(ConvParams &p, int B, bool use_classes, hipStream_t s)
{
    build_classes(p, B, BM, use_classes);
    p.ntiles = mpsr::ceil_div(p.N, BN);
    p.fd_ntiles = make_fastdiv(p.ntiles);
    p.fd_mtiles = make_fastdiv(p.mtiles_xcd);
    const long long blocks = 8LL * p.mtiles_xcd * p.ntiles * p.splits;
    if (blocks > 0x7fffffffLL) return mpsr::fail(MPSR_ERR_UNSUPPORTED, "conv2d: grid too large");
    // 1x1 layers over whole images with whole 32-channel K steps take the instantiation without decode / tap logic
    const bool plain = p.KH == 1 && p.KW == 1 && p.C % BK == 0 && p.ncls == 1 && g_plain_override.load() != 0;
    // 3x3 layers whose classes cover both axes (every visited tap in-image for every row of its class); the moved
    // descriptor base must stay a valid 32-bit offset range
    const bool cls = use_classes && p.KH == 3 && p.KW == 3 && p.C % BK == 0 &&
                     p.H >= 2 * p.dil && p.W >= 2 * p.dil && g_plain_override.load() != 0 &&
                     (long long)p.xbytes + (long long)p.dil * (p.W + 1) * p.C * 4 < 0x7ffffff0LL;
    {
        if (plain) {
            hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, MATH, DEPTH, 1>), dim3((unsigned)blocks), dim3(256),
                               0, s, p);
            MPSR_CHECK_LAUNCH("conv_igemm_kernel");
            return MPSR_OK;
        }
        if (cls) {
            hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, MATH, DEPTH, 2>), dim3((unsigned)blocks), dim3(256),
                               0, s, p);
            MPSR_CHECK_LAUNCH("conv_igemm_kernel");
            return MPSR_OK;
        }
    }
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, MATH, DEPTH>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    MPSR_CHECK_LAUNCH("conv_igemm_kernel");
    return MPSR_OK;
}

// Resident workgroups of a stream-K instantiation: CUs x workgroups per CU, from the occupancy query, once per device
// and instantiation.  The query can be one per CU too high for SGPR-heavy kernels (MI355X_MICROARCH.md); the kernel
// never waits on another workgroup, so an over-estimate only costs balance, not correctness.
struct SkSlots {
    std::once_flag once[16];
    int slots[16] = {0};
};

template <int BM, int BN, int WM, int WN, int MATH>
int sk_slots(int &out)
{
    static SkSlots cache;
    int dev = 0;
    MPSR_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) dev = 0;
    hipError_t err = hipSuccess;
    std::call_once(cache.once[dev], [&]() {
        int per_cu = 0, cus = 0;
        err = hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &per_cu, reinterpret_cast<const void *>(conv_sk_kernel<BM, BN, WM, WN, MATH>), 256, 0);
        if (err == hipSuccess) err = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (err == hipSuccess && per_cu > 0 && cus > 0) cache.slots[dev] = per_cu * cus;
    });
    if (err != hipSuccess) return mpsr::fail(MPSR_ERR_HIP, "stream-K occupancy query: %s", hipGetErrorString(err));
    out = cache.slots[dev] > 0 ? cache.slots[dev] : 256 * 4;
    return MPSR_OK;
}

// tuning knobs (tests sweep them): tile -1 = heuristic, 0=128x128 1=128x64 2=64x128 3=64x64 4=128x32 5=96x128;
// classes -1 = heuristic, 0 = never, 1 = whenever the geometry allows;
// sched -1 = heuristic, 0 = one tile per workgroup, 1 = stream-K;
// sk_per_cu > 0 overrides the resident-workgroups-per-CU figure of the occupancy query, < 0 sets the workgroup count
std::atomic<int> g_tile_override{-1};
std::atomic<int> g_class_override{-1};
std::atomic<int> g_sched_override{-1};
std::atomic<int> g_sk_per_cu{0};
std::atomic<int> g_wino_policy{0};      // mpsr_set_winograd_policy: MPSR_WINOGRAD_AUTO / MPSR_WINOGRAD_OFF
std::atomic<int> g_wino_override{-1};   // Winograd for eligible 3x3 layers: -1 heuristic, 0 never, 1 F(2x2,3x3) / 2 F(4x4,3x3) / 3 F(3x3,3x3) on atrous sub-grids whenever possible
std::atomic<int> g_depth_override{-1};  // staging depth of the one-tile-per-workgroup kernel: -1 heuristic, 1, 2
std::atomic<int> g_math{MATH_FP32};  // mpsr_set_conv_math (the default of calls that do not say otherwise)
inline int cur_math() { return mpsr::t_call_math >= 0 ? mpsr::t_call_math : g_math.load(); }
inline int cur_wino_policy() { return mpsr::t_call_wino_policy >= 0 ? mpsr::t_call_wino_policy : g_wino_policy.load(); }
std::atomic<int> g_atrous_wino2{1};  // mpsr_debug_set_atrous_wino2: F(2x2,3x3) on the sub-grids of atrous layers in the automatic rule
std::atomic<int> g_atrous_wino2_min_pixels{12000};  // (two 40 x 152 maps; 288 vs 459 us at eight; the 256-channel layers from ONE map on: conv2d_winograd_choice)
std::atomic<int> g_wino3_halo{1};    // mpsr_debug_set_wino3_halo: the tiled F(3x3,3x3) form (block2's atrous layers) in the automatic rule

// Scratch a stream-K launch needs behind `ws`: two partial-tile slabs per workgroup, then one counter per tile.
inline size_t sk_scratch_floats(int G, int BM, int BN, long long tiles)
{
    return (size_t)G * 2 * BM * BN + (size_t)tiles;
}

// Returns MPSR_OK and sets `done` when the layer was launched in stream-K form; leaves `done` false when the scratch
// is too small (the caller then takes the one-tile-per-workgroup kernel).
template <int BM, int BN, int WM, int WN, int MATH = MATH_FP32>
int launch_sk(ConvParams &p, int B, bool use_classes, int mode, bool counters_clean, hipStream_t s, bool &done)
{
    done = false;
    build_classes(p, B, BM, use_classes);
    p.ntiles = mpsr::ceil_div(p.N, BN);
    p.splits = 1;
    SkParams sk;
    unsigned long long U = 0, T = 0;
    for (int c = 0; c < p.ncls; ++c) {
        const PixelClass &pc = p.cls[c];
        sk.ks[c] = (pc.ky1 - pc.ky0 + 1) * (pc.kx1 - pc.kx0 + 1) * p.cblocks;
        sk.ubase[c] = (unsigned)U;
        sk.tbase[c] = (unsigned)T;
        const unsigned long long tiles = (unsigned long long)pc.tiles * p.ntiles;
        T += tiles;
        U += tiles * sk.ks[c];
    }
    for (int c = p.ncls; c <= MAX_CLS; ++c) {
        sk.ubase[c] = (unsigned)U;
        sk.tbase[c] = (unsigned)T;
        if (c < MAX_CLS) sk.ks[c] = 1;
    }
    if (U == 0 || U > 0x7fffffffULL || T > 0x7fffffffULL) return MPSR_OK;
    int slots = 0;
    if (int rc = sk_slots<BM, BN, WM, WN, MATH>(slots)) return rc;
    const int per_cu = g_sk_per_cu.load();  // tuning: > 0 workgroups per CU, < 0 absolute workgroup count
    if (per_cu > 0) {
        int dev = 0, cus = 256;
        MPSR_CHECK_HIP(hipGetDevice(&dev));
        MPSR_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        slots = per_cu * cus;
    } else if (per_cu < 0) {
        slots = -per_cu;
    }
    // every workgroup should keep at least ~4 K steps; G a multiple of 8 (one eighth of the sequence per XCD)
    long long G = slots;
    if (G > (long long)(U / 4)) G = (long long)(U / 4);
    G = G / 8 * 8;
    if (G < 8) G = 8;
    while (G > 8 && sk_scratch_floats((int)G, BM, BN, (long long)T) > (p.ws ? p.ws_floats : 0)) G -= 8;
    if (!p.ws || sk_scratch_floats((int)G, BM, BN, (long long)T) > p.ws_floats) return MPSR_OK;
    sk.G = (int)G;
    sk.U = (unsigned)U;
    sk.q = (unsigned)(U / G);
    sk.r = (unsigned)(U % G);
    sk.slabs = p.ws;
    sk.counters = reinterpret_cast<int *>(p.ws + (size_t)G * 2 * BM * BN);
    if (!counters_clean) MPSR_CHECK_HIP(hipMemsetAsync(sk.counters, 0, sizeof(int) * (size_t)T, s));
    hipLaunchKernelGGL((conv_sk_kernel<BM, BN, WM, WN, MATH>), dim3((unsigned)G), dim3(256), 0, s, p, sk);
    MPSR_CHECK_LAUNCH("conv_sk_kernel");
    done = true;
    return MPSR_OK;
}

}  // namespace

namespace mpsr {

int conv3x3_narrow(const float *x, int B, int H, int W, int C, const float *w, const float *bias, int relu, float *y,
                   int N, hipStream_t s, int in_c8);  // image_ops.hip
// winograd.hip
size_t winograd_scratch_floats(int C, int N);
bool winograd_applies(int H, int W, int C, int N);
bool winograd_applies_dilated(int H, int W, int C, int N, int dilation);
int conv3x3_winograd(const float *x, int B, int H, int W, int C, const float *w, const float *bias, int relu, float *y,
                     int N, float *ws, size_t ws_floats, hipStream_t s, int dilation);
// winograd4.hip
size_t winograd4_scratch_floats(int C, int N);
bool winograd4_applies(int H, int W, int C, int N);
int conv3x3_winograd4(const float *x, int B, int H, int W, int C, const float *w, const float *bias, int relu,
                      float *y, int N, float *ws, size_t ws_floats, hipStream_t s, int in_c8, int out_c8, float *part,
                      size_t part_floats);
size_t winograd4_split_floats(int B, int H, int W, int N);
extern std::atomic<int> g_wino4_split;
// winograd3.hip
size_t winograd3_scratch_floats(int C, int N);
bool winograd3_applies(int H, int W, int C, int dilation);
double winograd3_executed_flops(int B, int H, int C, int N, int dilation);
int winograd3_form(int B, int H, int W, int C, int N, int dilation);
int conv3x3_winograd3(const float *x, int B, int H, int W, int C, const float *w, const float *bias, int relu,
                      float *y, int N, int dilation, float *ws, size_t ws_floats, hipStream_t s,
                      const float *mask = nullptr);

// pointwise.hip
bool pointwise_applies(long long M, int K, int N);
int pointwise_override();
int conv1x1_pointwise(const float *x, long long M, int K, const float *w, const float *bias, const float *residual,
                      int relu, float *y, int N, hipStream_t s);

// thin_conv.hip
bool thin_input_conv_applies(int B, int H, int W, int C, int N, int KH, int KW, int dilation);
int thin_input_conv(const float *x, int B, int H, int W, const float *w, const float *bias, int relu, float *y, int N,
                    hipStream_t s);

bool fc_rows_applies(long long M, int K, int N);
bool fc_rows_split_applies(long long M, int K, int N, const float *bias, const float *y, const float *ws, size_t ws_floats);
int fc_rows_split(const float *x, long long M, int K, const float *w, const float *bias, int relu, float *y, int N,
                  float *ws, hipStream_t s);
int fc_rows(const float *x, long long M, int K, const float *w, const float *bias, const float *residual, int relu,
            float *y, int N, hipStream_t s);

// True when conv2d() sends a fully-connected layer (H = W = 1: a row is an instance) with few rows to fc_rows_kernel
// (pointwise.hip): the heads' layers, whose 64 x 64 tiles would cover a quarter of the CUs.  The rule looks at nothing
// but "FC with at most 2048 rows": that kernel sums K in another order than the implicit GEMM, and an instance's
// result must not depend on how a batch was cut into chunks (DeviceNet.MAX_CHUNK; test_chunked_batches_*).
bool conv2d_takes_fc_rows(int H, int W, long long M, int C, int N, int KH, int KW, int split_k)
{
    const int pw = pointwise_override();
    return H == 1 && W == 1 && KH == 1 && KW == 1 && split_k <= 1 && pw != 0 && cur_math() == MATH_FP32 &&
           g_tile_override.load() < 0 && g_sched_override.load() < 0 && fc_rows_applies(M, C, N);
}

// True when conv2d() sends a 1x1 layer to the persistent pointwise kernel (pointwise.hip): the wide trunk layers, where
// the launch has at least two 96 x 128 tiles per CU (measured: block3's conv1 / conv3 / shortcut, block2's conv3, the
// squash layers; block2's conv1 with its 384 tiles stays on the implicit GEMM).
bool conv2d_takes_pointwise(long long M, int C, int N, int KH, int KW, int split_k)
{
    const int pw = pointwise_override();
    if (KH != 1 || KW != 1 || split_k > 1 || pw == 0 || cur_math() != MATH_FP32 || g_tile_override.load() >= 0 ||
        !pointwise_applies(M, C, N))
        return false;
    if (pw > 0) return true;
    // (a forced schedule / border-class mode asks for the implicit-GEMM kernels, as for the few-row FC kernel above)
    if (g_sched_override.load() >= 0 || g_class_override.load() >= 0) return false;
    const long long tiles = ((M + 95) / 96) * ((N + 127) / 128);
    return split_k == 0 && N % 128 == 0 && tiles >= 512;
}

// Scratch behind `ws` when the caller leaves the schedule to the library (split_k == 0).
//  * stream-K: two partial-tile slabs per persistent workgroup + one counter per tile.  Bounded over every tile
//    instantiation by 8 workgroups/CU x 64x64 or 4/CU x 96x128 / 128x64 or 3/CU x 128x128 on 256 CUs, i.e.
//    1024 x 2 x 96 x 128 floats, plus (ceil(M / 64) + 9) * ceil(N / 32) counters (smallest tile, 9 pixel classes);
//  * the older automatic split-K of the one-tile-per-workgroup kernel (kept for callers whose scratch is smaller):
//    a launch whose 64x64 tiles would leave most CUs with one workgroup or none is cut along K so that ~3 workgroups
//    per CU exist, each keeping at least 4 K steps: (768 + tiles) * 64 * 64 floats with tiles < 768.
size_t conv_auto_split_floats() { return (size_t)1536 * 64 * 64; }
size_t conv_scratch_floats(long long M, int N)
{
    const size_t slabs = (size_t)1024 * 2 * 96 * 128;
    const size_t counters = (size_t)((M + 63) / 64 + 9) * (size_t)((N + 31) / 32);
    return slabs + counters;
}

// A launch with at least g_auto_split_enough tiles is left alone; a smaller one is cut along K into enough slices for
// g_auto_split_target workgroups (mpsr_debug_set_auto_split_target).  r06 sweep of the aim (tools/full_path_bench.py
// --knob ..., tools/step_knob_run.py; profiles/r06_step_ab.txt): 1152 takes the one-image path from 5.63 to 5.48 ms (the
// full-image trunk's 380-tile layers in 4 slices instead of 3) and costs the crop trunk alone +1.7 % at 24 boxes and
// +0.8 % at 48; 1536 loses on both.  Not enough of a pattern to move the default.
int g_auto_split_target = 768;
constexpr int g_auto_split_enough = 768;
constexpr int g_auto_split_min_ksteps = 4;

int auto_split_k(int M, int N, int ksteps, const float *ws, size_t ws_floats)
{
    if (!ws) return 1;
    const long long tiles = (long long)ceil_div(M, 64) * ceil_div(N, 64);
    if (tiles >= g_auto_split_enough || tiles >= g_auto_split_target || N <= 32) return 1;
    long long s = (g_auto_split_target + tiles - 1) / tiles;
    if (s > 8) s = 8;
    if (s > ksteps / g_auto_split_min_ksteps) s = ksteps / g_auto_split_min_ksteps;
    while (s > 1 && (size_t)s * M * N > ws_floats) --s;
    return s < 1 ? 1 : (int)s;
}

// The Winograd selector the rules below read: the debug override if set, else "never" under MPSR_WINOGRAD_OFF.
static int winograd_mode()
{
    const int wino = g_wino_override.load();
    return (wino < 0 && cur_wino_policy() == MPSR_WINOGRAD_OFF) ? 0 : wino;
}
// MPSR_WINOGRAD_ACCURATE (r06): of the transform-domain forms only those that keep an element-wise 1e-3 on heavy-tailed
// maps -- the sixteen-product tiles and F(2x2,3x3); no F(4x4,3x3), no F(3x3,3x3) tiles with halos.  (A debug override of
// the Winograd selector wins, as everywhere.)
static bool accurate_only() { return g_wino_override.load() < 0 && cur_wino_policy() == MPSR_WINOGRAD_ACCURATE; }

// True when conv2d() with split_k = 0 would send this 3x3 layer to the F(4x4,3x3) kernel (network.hip asks before it
// lays the decoder's internal tensors out channel-blocked, which only that kernel reads).
bool conv2d_takes_winograd4(int B, int H, int W, int C, int N, const float *ws, size_t ws_floats)
{
    const long long M64 = (long long)B * H * W;
    const int wino = winograd_mode();
    if (accurate_only()) return false;
    return ws && (wino < 0 || wino == 2) && g_tile_override.load() < 0 && M64 >= 65536 && C >= 64 && N >= 64 &&
           winograd4_applies(H, W, C, N) && ws_floats >= winograd4_scratch_floats(C, N) && M64 * C * 4 < 0x7f000000LL;
}

// True when conv2d() sends a (residual-free) atrous 3x3 layer to the F(3x3,3x3) kernel (network.hip asks before it
// transforms the filters of all such layers of a trunk in one launch).
bool conv2d_takes_winograd3(int B, int H, int W, int C, int N, int KH, int KW, int dilation, int split_k, const float *ws,
                            size_t ws_floats)
{
    const int wino = winograd_mode();
    const long long M64 = (long long)B * H * W;
    const bool can3 = KH == 3 && KW == 3 && split_k == 0 && ws && cur_math() == MATH_FP32 &&
                      winograd3_applies(H, W, C, dilation) && ws_floats >= winograd3_scratch_floats(C, N) &&
                      M64 * C * 4 < 0x7f000000LL;
    // (tiles of 3x3 outputs: th x th per pixel sub-grid; the automatic rule takes the forms that were measured -- one
    // tile per sub-grid (block3, 12x12 at dilation 4), 2 x 2 tiles with halos (block2, 12x12 at dilation 2) and the dense
    // 3x3 layers of 12x12 maps as 4 x 4 tiles (block1: too few 4x4 blocks for the F(4x4) kernel's workgroups))
    const int th = can3 ? H / (3 * dilation) : 1;
    // ... and, at batches too small for the F(4x4) kernel's 32-tile workgroups to fill the chip (the reference's own
    // 32 boxes per image: its rule wants 65536 pixels), the decoder's dense 3x3 layers as up to 16 x 16 tiles of 3x3:
    // 204 -> 79 us (24x24x256, B = 32), 115 -> 83 us (48x48x128)
    // enough tiles to fill the chip -- except where the sixteen-product kernel serves the layer (one tile per sub-grid):
    // it cuts small launches along K (winograd3z.hip, SPLIT) and is ahead of the implicit GEMM at every batch measured
    // (29-32 us against 45-48 at 4 .. 32 instances, 38 against 72 at 48)
    const long long min_tiles = (can3 && th == 1 && winograd3_form(B, H, W, C, N, dilation) == 2) ? 1 : 1024;
    const int th_max = !g_wino3_halo.load() ? 1 : dilation > 1 ? 2 : conv2d_takes_winograd4(B, H, W, C, N, ws, ws_floats) ? 4 : 16;
    const bool want3 = wino == 3 || (wino < 0 && g_tile_override.load() < 0 && g_class_override.load() < 0 &&
                                     th <= th_max && (dilation > 1 || th > 1) &&
                                     (long long)B * dilation * dilation * th * th >= min_tiles && C >= 64 && N >= 64);
    if (accurate_only() && !(th == 1 && winograd3_form(B, H, W, C, N, dilation) == 2)) return false;
    return can3 && want3;
}

// The dense-3x3 choice of conv2d() -- 0: no Winograd kernel, 1: F(2x2,3x3) (winograd.hip), 2: F(4x4,3x3) (winograd4.hip).
// One copy of the rule: mpsr_conv2d_plan asks it too.
int conv2d_winograd_choice(int B, int H, int W, int C, int N, int KH, int KW, int dilation, bool residual, int split_k,
                           const float *ws, size_t ws_floats)
{
    const long long M64 = (long long)B * H * W;
    int wino = winograd_mode();
    // (the F(4x4) kernel also serves the opt-in bf16x3 mode: in exact fp32 it is faster on these layers than the
    // split-bfloat16 implicit GEMM -- 3.1 vs 3.6 ms for the four of them -- and adds no drift)
    const bool base = KH == 3 && KW == 3 && dilation == 1 && !residual && split_k <= 1 && ws;
    // F(2x2,3x3) also serves atrous layers whose pixel sub-grids are whole and even-sized (the FULL-image trunk's block2 /
    // block3: 40 x 152 at dilation 2 / 4 = sub-grids of 20 x 76 / 10 x 38; the crop trunk's 12 x 12 maps were taken by
    // the F(3x3,3x3) kernels before this rule is asked): 16 products per 2x2 block where the border-class implicit GEMM
    // executes ~34
    const bool atrous2 = KH == 3 && KW == 3 && dilation > 1 && !residual && split_k <= 1 && ws &&
                         winograd_applies_dilated(H, W, C, N, dilation) && M64 * C * 4 < 0x7f000000LL;
    const bool can2 = (base || atrous2) && cur_math() == MATH_FP32 && winograd_applies_dilated(H, W, C, N, dilation) &&
                      ws_floats >= winograd_scratch_floats(C, N);
    const bool can4 = base && winograd4_applies(H, W, C, N) && ws_floats >= winograd4_scratch_floats(C, N) &&
                      M64 * C * 4 < 0x7f000000LL;
    // (one 40 x 152 map = 6080 pixels = 96 workgroups: the kernel cuts such launches along K -- winograd.hip, SPLIT --
    // and is then ahead on the 256-channel layers, 61 vs 94 us; the 128-channel ones are equal, 35 us, and stay)
    const long long atrous2_min = C >= 256 ? std::min(6000, g_atrous_wino2_min_pixels.load()) : g_atrous_wino2_min_pixels.load();
    // (the automatic choice is conv2d_takes_winograd4's -- network.hip asks it too)
    if (wino < 0) wino = split_k != 0 ? 0 : (base && conv2d_takes_winograd4(B, H, W, C, N, ws, ws_floats)) ? 2
                         : (base && M64 >= 65536 && C >= 64 && N >= 64 && g_tile_override.load() < 0) ? 1
                         : (atrous2 && g_atrous_wino2.load() && M64 >= atrous2_min && C >= 64 && N >= 64 &&
                            g_tile_override.load() < 0 && g_class_override.load() < 0) ? 1 : 0;
    if (wino == 2 && can4) return 2;
    if ((wino == 1 || wino == 2) && can2) return 1;
    return 0;
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
        g_tile_override.load() < 0 && ((uintptr_t)w & 3) == 0)
        return conv3x3_narrow(x, B, H, W, C, w, bias, relu, y, N, stream, 0);
    // ... and 3x3 layers with FOUR input channels (the data gradient of that head: dy padded to 4 channels -> 128) are
    // output-bandwidth-bound: 36 multiply-adds per output element on the vector ALU (thin_conv.hip)
    if (!residual && split_k <= 1 && g_tile_override.load() < 0 && cur_math() == MATH_FP32 &&
        thin_input_conv_applies(B, H, W, C, N, KH, KW, dilation))
        return thin_input_conv(x, B, H, W, w, bias, relu, y, N, stream);
    // atrous 3x3 layers whose pixel sub-grids are 3x3 (block3's conv2: 12x12 at dilation 4): every sub-grid is one
    // Winograd F(3x3,3x3) tile with an all-zero halo -- 25 products where the border-class implicit GEMM executes 49
    // (winograd3.hip).  fp32 mode only (the bf16x3 implicit GEMM is faster than fp32 Winograd there).
    if (!residual && conv2d_takes_winograd3(B, H, W, C, N, KH, KW, dilation, split_k, ws, ws_floats))
        return conv3x3_winograd3(x, B, H, W, C, w, bias, relu, y, N, dilation, ws, ws_floats, stream);
    // the big dense 3x3 layers (map decoder) go to a Winograd kernel when the caller leaves the schedule to the
    // library: F(4x4,3x3) (winograd4.hip, 4x fewer multiply-adds) where the map divides into 4x4 blocks, else
    // F(2x2,3x3) (winograd.hip, 2.25x fewer)
    {
        const int wc = conv2d_winograd_choice(B, H, W, C, N, KH, KW, dilation, residual != nullptr, split_k, ws, ws_floats);
        if (wc == 2) return conv3x3_winograd4(x, B, H, W, C, w, bias, relu, y, N, ws, ws_floats, stream, 0, 0, nullptr, 0);
        if (wc == 1) return conv3x3_winograd(x, B, H, W, C, w, bias, relu, y, N, ws, ws_floats, stream, dilation);
    }
    if (conv2d_takes_pointwise(M64, C, N, KH, KW, split_k))
        return conv1x1_pointwise(x, M64, C, w, bias, residual, relu, y, N, stream);
    if (conv2d_takes_fc_rows(H, W, M64, C, N, KH, KW, split_k))
        return fc_rows(x, M64, C, w, bias, residual, relu, y, N, stream);
    // ... and with a K too long for that kernel's single workgroup per tile (img_fc, K = 18432) at few rows: its K-split form
    if (H == 1 && W == 1 && KH == 1 && KW == 1 && !residual && split_k == 0 && pointwise_override() != 0 &&
        cur_math() == MATH_FP32 && g_tile_override.load() < 0 && g_sched_override.load() < 0 &&
        fc_rows_split_applies(M64, C, N, bias, y, ws, ws_floats))
        return fc_rows_split(x, M64, C, w, bias, relu, y, N, ws, stream);
    ConvParams p;
    p.x = x; p.w = w; p.bias = bias; p.residual = residual; p.y = y; p.ws = ws; p.ws_floats = ws ? ws_floats : 0;
    p.M = (int)M64; p.H = H; p.W = W; p.C = C; p.N = N; p.KH = KH; p.KW = KW; p.dil = dilation; p.relu = relu;
    p.xbytes = (unsigned)(M64 * C * 4);
    p.wbytes = (unsigned)((long long)N * KH * KW * C * 4);
    p.cblocks = ceil_div(C, BK);
    p.ksteps_total = KH * KW * p.cblocks;
    // border-class tiling: atrous 3x3 layers only (a 1-pixel border ring at dilation 1 saves too little)
    bool use_classes = KH == 3 && KW == 3 && dilation > 1;
    const int class_override = g_class_override.load(), tile_override = g_tile_override.load();
    const int math = cur_math();
    if (class_override == 0) use_classes = false;
    if (class_override == 1) use_classes = KH == 3 && KW == 3;
    int sel = tile_override;
    if (sel < 0) {
        // measured on MI355X (tools/conv_layer_bench.py --rounds, profiles/): 128x128 wins once there are >= ~18 tiles
        // per CU (the 24x24 / 48x48 decoder layers, 130-138 TFLOP/s); the 12x12 trunk layers (M = 36864 at B = 256)
        // only make 2.25 tiles of 128x128 per CU, so 64x64 tiles (9 per CU, 7 waves/SIMD) balance and overlap
        // better; the wide 1x1 layers among them (block3 conv1 / shortcut, squash: C >= 512, N a multiple of 128)
        // take 96x128 tiles -- 3 per CU, all resident at once, 40 % less L2 -> LDS traffic per multiply-add
        if (N <= 32) sel = 4;
        else if (p.M >= 131072) sel = N <= 64 ? 1 : 0;  // (N = 64: the root conv over im2col rows, 38 vs 66 us)
        else if (KH == 1 && KW == 1 && C >= 512 && N >= 256 && N % 128 == 0 && p.M >= 24576 && math == MATH_FP32) sel = 5;
        else sel = 3;
        // bf16x3: 3 bf16 matrix instructions replace 8 fp32 ones at 1/16 of the cycles each, so LDS and the operand
        // split feed the matrix pipes: larger wave tiles (fewer fragment reads per instruction) win -- 128x128 on the
        // big decoder maps, 128x64 on the 12x12 trunk layers (tools/conv_layer_bench.py --math bf16x3); small
        // batches -- a single image's 32 crops -- need the small tile to have workgroups for every CU
        if (math == MATH_BF16X3) sel = N <= 32 ? 4 : (p.M < 16384 ? 3 : ((N <= 64 || p.M < 131072) ? 1 : 0));
    }
    // schedule: stream-K only where it measured faster than the alternatives -- fully-connected layers with a very
    // long K and few rows (img_fc: 256 x 18432 x 2048: 216 us against 358 us for sliced K + a reduce kernel); on the
    // convolution layers its partial-tile hand-offs cost more than the balance buys
    int sched = g_sched_override.load();
    if (sched < 0) sched = (split_k == 0 && ws && p.M <= 2048 && p.ksteps_total >= 128) ? 1 : 0;
    if (sched == 1 && split_k <= 1 && ws) {
        bool done = false;
        int rc = MPSR_OK;
#define MPSR_SK(BM_, BN_, WM_, WN_)                                                                               \
    rc = math == MATH_BF16X3 ? launch_sk<BM_, BN_, WM_, WN_, MATH_BF16X3>(p, B, use_classes, sched, false, stream, done) \
                             : launch_sk<BM_, BN_, WM_, WN_, MATH_FP32>(p, B, use_classes, sched, false, stream, done)
        switch (sel) {
            case 0: MPSR_SK(128, 128, 2, 2); break;
            case 1: MPSR_SK(128, 64, 2, 2); break;
            case 2: MPSR_SK(64, 128, 2, 2); break;
            case 3: MPSR_SK(64, 64, 2, 2); break;
            case 5: MPSR_SK(96, 128, 1, 4); break;
            default: MPSR_SK(128, 32, 4, 1); break;
        }
#undef MPSR_SK
        if (rc) return rc;
        if (done) return MPSR_OK;
    }
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
    int rc;
    // two staging register sets (loads two K steps ahead) pay on the atrous 3x3 layers of the trunk (block3 conv2:
    // 228 -> 213 us) and with the 96x128 tile at N = 256; elsewhere the lost occupancy costs as much as it buys
    int depth = g_depth_override.load();
    // 1x1 layers on the 64x64 tile: the decode-free instantiation has registers to spare, two staging sets keep its
    // 7 workgroups per CU (measured +2 % on block2 / block3 conv3 and conv1)
    const bool plain64 = sel == 3 && KH == 1 && KW == 1 && C % BK == 0 && p.ksteps_total >= 8 && math == MATH_FP32;
    if (depth < 0) depth = ((sel == 3 && use_classes) || (sel == 5 && N <= 256) || plain64) ? 2 : 1;
#define MPSR_TILE(BM_, BN_, WM_, WN_)                                                                        \
    rc = math == MATH_BF16X3 ? launch<BM_, BN_, WM_, WN_, MATH_BF16X3>(p, B, use_classes, stream)           \
         : depth == 2        ? launch<BM_, BN_, WM_, WN_, MATH_FP32, 2>(p, B, use_classes, stream)          \
                             : launch<BM_, BN_, WM_, WN_, MATH_FP32, 1>(p, B, use_classes, stream)
    switch (sel) {
        case 0: MPSR_TILE(128, 128, 2, 2); break;
        case 1: MPSR_TILE(128, 64, 2, 2); break;
        case 2: MPSR_TILE(64, 128, 2, 2); break;
        case 3: MPSR_TILE(64, 64, 2, 2); break;
        case 5: MPSR_TILE(96, 128, 1, 4); break;
        default: MPSR_TILE(128, 32, 4, 1); break;
    }
#undef MPSR_TILE
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
extern "C" void mpsr_debug_set_auto_split_target(int workgroups) { mpsr::g_auto_split_target = workgroups; }
extern "C" void mpsr_debug_set_conv_tile(int sel) { g_tile_override = sel; }
extern "C" void mpsr_debug_set_conv_classes(int mode) { g_class_override = mode; }
extern "C" void mpsr_debug_set_conv_depth(int depth) { g_depth_override = depth; }
extern "C" void mpsr_debug_set_conv_plain(int mode) { g_plain_override = mode; }
extern "C" void mpsr_debug_set_conv_winograd(int mode) { g_wino_override = mode; }
extern "C" int mpsr_set_winograd_policy(int policy)
{
    MPSR_REQUIRE(policy == MPSR_WINOGRAD_AUTO || policy == MPSR_WINOGRAD_OFF || policy == MPSR_WINOGRAD_ACCURATE,
                 "set_winograd_policy: unknown policy %d", policy);
    g_wino_policy = policy;
    return MPSR_OK;
}
extern "C" int mpsr_get_winograd_policy(void) { return g_wino_policy; }
extern "C" void mpsr_debug_set_conv_sched(int mode, int per_cu)
{
    g_sched_override = mode;
    g_sk_per_cu = per_cu;
}

extern "C" int mpsr_set_conv_math(int mode)
{
    MPSR_REQUIRE(mode == MATH_FP32 || mode == MATH_BF16X3, "set_conv_math: unknown mode %d", mode);
    g_math = mode;
    return MPSR_OK;
}
extern "C" int mpsr_get_conv_math(void) { return g_math; }
extern "C" void mpsr_debug_set_wino3_halo(int on) { g_wino3_halo = on; }
extern "C" void mpsr_debug_set_atrous_wino2(int on, int min_pixels)
{
    g_atrous_wino2 = on;
    if (min_pixels > 0) g_atrous_wino2_min_pixels = min_pixels;
}

// What mpsr_conv2d_nhwc_f32 does with a layer when the schedule is left to it (split_k = 0, scratch provided, fp32):
// kind 0 = implicit GEMM (conv_igemm_kernel / conv_sk_kernel), 1 = Winograd F(2x2,3x3), 2 = direct narrow kernel,
// 3 = Winograd F(4x4,3x3), 4 = Winograd F(3x3,3x3) on the 3x3 sub-grids of an atrous layer, 5 = the pointwise kernel
// of the wide 1x1 layers, 6 = the few-row FC kernel (both pointwise.hip); and
// the multiply-add FLOPs the chosen kernel really issues (2 x MACs): the Winograd kernels 16/36 or 36/144 of the direct count,
// the implicit GEMM with border classes only the in-image taps.  For reporting (bench.py), not part of the compute path.
extern "C" int mpsr_conv2d_plan(int B, int H, int W, int C, int N, int KH, int KW, int dilation, int *kind,
                                double *executed_flops)
{
    MPSR_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && N > 0 && kind && executed_flops, "conv2d_plan: bad arguments");
    const double M = (double)B * H * W;
    // (the same predicate conv2d() asks; the plan assumes the scratch is there)
    if (mpsr::conv2d_takes_winograd3(B, H, W, C, N, KH, KW, dilation, 0, reinterpret_cast<const float *>(16), ~(size_t)0)) {
        *kind = 4;  // 25 products per 3x3 tile (th x th tiles per pixel sub-grid: the launcher's own count)
        *executed_flops = mpsr::winograd3_executed_flops(B, H, C, N, dilation);
        return MPSR_OK;
    }
    const int wc = mpsr::conv2d_winograd_choice(B, H, W, C, N, KH, KW, dilation, false, 0,
                                                reinterpret_cast<const float *>(16), ~(size_t)0);
    if (wc == 2) {
        *kind = 3;  // F(4x4,3x3): 36 products per 4x4 block
        *executed_flops = 2.0 * (double)B * (H / 4) * (W / 4) * 36.0 * C * N;
        return MPSR_OK;
    }
    if (wc == 1) {
        *kind = 1;
        *executed_flops = 2.0 * (double)B * (H / 2) * (W / 2) * 16.0 * C * N;
        return MPSR_OK;
    }
    if (KH == 3 && KW == 3 && dilation == 1 && N <= 4 && C % 32 == 0) {
        *kind = 2;
        *executed_flops = 2.0 * M * 9.0 * C * N;
        return MPSR_OK;
    }
    if (mpsr::thin_input_conv_applies(B, H, W, C, N, KH, KW, dilation)) {
        *kind = 2;  // (the same family: a direct vector-ALU kernel, here with the thin side on the input)
        *executed_flops = 2.0 * M * 9.0 * C * N;
        return MPSR_OK;
    }
    if (mpsr::conv2d_takes_pointwise((long long)B * H * W, C, N, KH, KW, 0)) {
        *kind = 5;  // the persistent pointwise kernel
        *executed_flops = 2.0 * M * C * N;
        return MPSR_OK;
    }
    if (mpsr::conv2d_takes_fc_rows(H, W, (long long)B * H * W, C, N, KH, KW, 0)) {
        *kind = 6;  // few-row FC kernel
        *executed_flops = 2.0 * M * C * N;
        return MPSR_OK;
    }
    *kind = 0;
    double taps = (double)KH * KW * H * W;  // tap evaluations per image and channel pair
    if (KH == 3 && KW == 3 && dilation > 1 && H >= 2 * dilation && W >= 2 * dilation) {
        // border classes: rows / columns within `dilation` of an edge lose one row / column of taps
        const double ty = 3.0 * H - 2.0 * dilation, tx = 3.0 * W - 2.0 * dilation;
        taps = ty * tx;
    }
    *executed_flops = 2.0 * (double)B * taps * C * N;
    return MPSR_OK;
}

extern "C" size_t mpsr_conv2d_scratch_floats(int B, int H, int W, int N)
{
    if (B <= 0 || H <= 0 || W <= 0 || N <= 0) return 0;
    size_t n = mpsr::conv_scratch_floats((long long)B * H * W, N);
    // with the (opt-in) position-split Winograd kernel enabled, large maps with N a multiple of 128 park the partial
    // outputs of half their workgroups behind the transformed filters (winograd4.hip)
    if (mpsr::g_wino4_split.load() != 0 && (long long)B * H * W >= 65536 && N % 128 == 0 && H % 4 == 0 && W % 4 == 0)
        n += mpsr::winograd4_split_floats(B, H, W, N) + 64;
    return n;
}

extern "C" int mpsr_conv2d_nhwc_f32(const float *x, int B, int H, int W, int C, const float *w, const float *bias,
                                    const float *residual, float *y, int N, int KH, int KW, int dilation, int relu,
                                    int split_k, float *ws, size_t ws_floats, mpsr_stream_t stream)
{
    return mpsr::conv2d(x, B, H, W, C, w, bias, residual, y, N, KH, KW, dilation, relu, split_k, ws, ws_floats,
                        mpsr::as_stream(stream));
}

// The same under per-call options: the arithmetic mode and the Winograd policy of THIS call, whatever the process-wide
// defaults say (the reference's launchers are stateless, tf_nndistance.cpp:168; SURVEY 8(b) "no state").
extern "C" int mpsr_conv2d_nhwc_f32_ex(const float *x, int B, int H, int W, int C, const float *w, const float *bias,
                                       const float *residual, float *y, int N, int KH, int KW, int dilation, int relu,
                                       int split_k, float *ws, size_t ws_floats, const mpsr_conv_opts *opts,
                                       mpsr_stream_t stream)
{
    if (opts) {
        MPSR_REQUIRE(opts->math >= 0 && opts->math <= MPSR_CALL_MATH_BF16X3, "conv2d: unknown opts.math %d", opts->math);
        MPSR_REQUIRE(opts->winograd_policy >= 0 && opts->winograd_policy <= MPSR_CALL_WINOGRAD_ACCURATE,
                     "conv2d: unknown opts.winograd_policy %d", opts->winograd_policy);
    }
    mpsr::CallOptsGuard guard(opts ? opts->math : 0, opts ? opts->winograd_policy : 0);
    return mpsr::conv2d(x, B, H, W, C, w, bias, residual, y, N, KH, KW, dilation, relu, split_k, ws, ws_floats,
                        mpsr::as_stream(stream));
}

// y = conv(x, w) where mask > 0, else 0 (mask shaped like y): a data gradient through the ReLU of the layer it belongs to.
// The F(3x3,3x3) atrous kernel applies the mask in its epilogue; every other shape runs the convolution as scheduled and
// then mpsr_relu_grad in place.
extern "C" int mpsr_conv2d_relu_masked_f32(const float *x, int B, int H, int W, int C, const float *w, const float *mask,
                                           float *y, int N, int KH, int KW, int dilation, float *ws, size_t ws_floats,
                                           mpsr_stream_t stream)
{
    MPSR_REQUIRE(mask, "conv2d_relu_masked: null mask");
    hipStream_t s = mpsr::as_stream(stream);
    if (B > 0 && x && w && y && mpsr::conv2d_takes_winograd3(B, H, W, C, N, KH, KW, dilation, 0, ws, ws_floats))
        return mpsr::conv3x3_winograd3(x, B, H, W, C, w, nullptr, 0, y, N, dilation, ws, ws_floats, s, mask);
    if (int rc = mpsr::conv2d(x, B, H, W, C, w, nullptr, nullptr, y, N, KH, KW, dilation, 0, 0, ws, ws_floats, s)) return rc;
    return mpsr_relu_grad(y, mask, y, (long long)B * H * W * N, stream);
}

// ------------------------------------------------------------------------------------------------ calibration
// What this box's matrix pipes sustain in fp32: nothing but v_mfma_f32_32x32x2_f32 on `chains` independent
// accumulators per wave (1 = one dependent chain, like a 32x32 wave tile; 4 = like a 64x64 wave tile).  MI355X boards
// differ in sustained clock under this load by up to ~20 % (power capping), so bench.py and the tuning tools quote
// kernel rates next to this figure measured in the same process, not only next to the 2.4 GHz datasheet peak.
namespace {
template <int CHAINS>
__global__ __launch_bounds__(256) void mfma_peak_kernel(float *out, int iters, float a0, float b0)
{
    f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16 / CHAINS; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[c][e];
    if (s == 12345.678f) out[0] = s;  // keeps the chain live; never true for the inputs used
}
}  // namespace

namespace {
// The K loop's instruction mix without its memory traffic: per step 8 ds_read_b128 feeding 16 dependent MFMAs, waits
// placed as hipcc places them in conv_igemm_kernel<64,64> (MODE 1), or reads issued but never waited for (MODE 0).
template <int MODE>
__global__ __launch_bounds__(256) void mfma_lds_kernel(float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[128 * LDS_STRIDE];
    for (int i = threadIdx.x; i < 128 * LDS_STRIDE; i += 256) lds[i] = 1e-3f * (float)(i & 15);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *Aw = lds + ((wave >> 1) * 32 + (lane & 31)) * LDS_STRIDE + (lane >> 5) * 4;
    const float *Bw = lds + (64 + (wave & 1) * 32 + (lane & 31)) * LDS_STRIDE + (lane >> 5) * 4;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    f32x4 keep = {1.f, 1.f, 1.f, 1.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 a = *reinterpret_cast<const f32x4 *>(Aw + kb * 8 + (i & 1) * 4);
            f32x4 b = *reinterpret_cast<const f32x4 *>(Bw + kb * 8 + (i & 1) * 4);
            if (MODE == 0) {
                asm volatile("" ::"v"(a), "v"(b));
                a = keep;
                b = keep;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[e];
    if (s == 12345.678f) out[0] = s;
}
}  // namespace

namespace {
// One dependent MFMA chain per wave with NV independent vector-ALU instructions issued after every MFMA: does ordinary
// VALU work of the resident waves take time away from the matrix pipe?  (tools/mfma_peak.py --valu)
template <int NV, int KIND = 0>
__global__ __launch_bounds__(256) void mfma_valu_kernel(float *out, int iters, float av, float bv)
{
    __shared__ __attribute__((aligned(16))) float buf[256 * 4 + 64];
    buf[threadIdx.x * 4] = av;
    __syncthreads();
    const float *lp = buf + (threadIdx.x & 63) * 4;
    __shared__ __attribute__((aligned(16))) float wbuf[256 * 4 + 2048];
    const unsigned wp = (unsigned)(size_t)(wbuf + threadIdx.x), wp2 = (unsigned)(size_t)(wbuf + threadIdx.x * 2),
                   wp4 = (unsigned)(size_t)(wbuf + threadIdx.x * 4);
    __attribute__((ext_vector_type(2))) float w2 = {av, bv};
    int sreg = iters;
    f32x4 lv = {0.f, 0.f, 0.f, 0.f};
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = av * (float)(threadIdx.x + e);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(u * NV + q) & 15]) : "v"(bv));
                if (KIND == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg));
                if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(lv) : "v"((unsigned)(size_t)lp) : "memory");
                if (KIND == 3) asm volatile("s_nop 0");
                // LDS stores of this thread's own slot (conflict-free): 4, 8, 16 bytes, and the paired 4-byte form
                if (KIND == 4) asm volatile("ds_write_b32 %0, %1" ::"v"(wp), "v"(v[q & 15]) : "memory");
                if (KIND == 5) asm volatile("ds_write_b64 %0, %1" ::"v"(wp2), "v"(w2) : "memory");
                if (KIND == 6) asm volatile("ds_write_b128 %0, %1" ::"v"(wp4), "v"(lv) : "memory");
                if (KIND == 7) asm volatile("ds_write2st64_b32 %0, %1, %2 offset1:4" ::"v"(wp), "v"(v[q & 15]), "v"(v[(q + 1) & 15]) : "memory");
            }
            if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[e] + v[e];
    s += lv.x + (float)sreg;
    if (s == 12345.678f) out[0] = s;
}
}  // namespace

namespace {
__global__ __launch_bounds__(256) void empty_kernel(float *out, int spin)
{
    extern __shared__ float dyn[];
    float v = 0.f;
    for (int i = 0; i < spin; ++i) asm volatile("s_sleep 1");
    if (out && threadIdx.x == 1000) out[0] = v + dyn[0];
}
}  // namespace

// Workgroup dispatch rate: `blocks` workgroups of 256 threads with `lds_bytes` of LDS that do nothing (spin = 0) or
// sleep for spin * 64 cycles.
extern "C" int mpsr_debug_dispatch(float *out, int blocks, int lds_bytes, int spin, mpsr_stream_t stream)
{
    MPSR_REQUIRE(blocks > 0 && lds_bytes >= 0 && lds_bytes <= 64 * 1024, "dispatch: bad arguments");
    hipLaunchKernelGGL(empty_kernel, dim3((unsigned)blocks), dim3(256), (size_t)lds_bytes, mpsr::as_stream(stream), out, spin);
    MPSR_CHECK_LAUNCH("empty_kernel");
    return MPSR_OK;
}

// kind 0: vector ALU, 1: scalar ALU, 2: LDS reads (ds_read_b128), 3: s_nop, 4-7: LDS stores (ds_write_b32 / _b64 /
// _b128 / ds_write2st64_b32); nv = 1, 2 (stores only), 4 or 8 of them after every MFMA
extern "C" int mpsr_debug_mfma_mix(float *out, int cus, int waves_per_simd, int nv, int kind, int iters,
                                   mpsr_stream_t stream)
{
    MPSR_REQUIRE(out && cus > 0 && waves_per_simd >= 1 && waves_per_simd <= 8 && iters > 0 &&
                     (nv == 4 || nv == 8 || ((nv == 1 || nv == 2) && kind >= 4)) && kind >= 0 && kind <= 7,
                 "mfma_mix: bad arguments");
    const dim3 grid((unsigned)(cus * waves_per_simd));
    hipStream_t s = mpsr::as_stream(stream);
#define MIX(NV_, K_) hipLaunchKernelGGL((mfma_valu_kernel<NV_, K_>), grid, dim3(256), 0, s, out, iters, 1.f, 1e-3f)
    if (kind >= 4) {
        if (nv == 1) { if (kind == 4) MIX(1, 4); else if (kind == 5) MIX(1, 5); else if (kind == 6) MIX(1, 6); else MIX(1, 7); }
        else if (nv == 2) { if (kind == 4) MIX(2, 4); else if (kind == 5) MIX(2, 5); else if (kind == 6) MIX(2, 6); else MIX(2, 7); }
        else if (nv == 4) { if (kind == 4) MIX(4, 4); else if (kind == 5) MIX(4, 5); else if (kind == 6) MIX(4, 6); else MIX(4, 7); }
        else { if (kind == 4) MIX(8, 4); else if (kind == 5) MIX(8, 5); else if (kind == 6) MIX(8, 6); else MIX(8, 7); }
    } else if (nv == 4) {
        if (kind == 0) MIX(4, 0); else if (kind == 1) MIX(4, 1); else if (kind == 2) MIX(4, 2); else MIX(4, 3);
    } else {
        if (kind == 0) MIX(8, 0); else if (kind == 1) MIX(8, 1); else if (kind == 2) MIX(8, 2); else MIX(8, 3);
    }
#undef MIX
    MPSR_CHECK_LAUNCH("mfma_valu_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_debug_mfma_valu(float *out, int cus, int waves_per_simd, int nv, int iters, mpsr_stream_t stream)
{
    MPSR_REQUIRE(out && cus > 0 && waves_per_simd >= 1 && waves_per_simd <= 8 && iters > 0, "mfma_valu: bad arguments");
    const dim3 grid((unsigned)(cus * waves_per_simd));
    hipStream_t s = mpsr::as_stream(stream);
    switch (nv) {
    case 0: hipLaunchKernelGGL(mfma_valu_kernel<0>, grid, dim3(256), 0, s, out, iters, 1.f, 1e-3f); break;
    case 1: hipLaunchKernelGGL(mfma_valu_kernel<1>, grid, dim3(256), 0, s, out, iters, 1.f, 1e-3f); break;
    case 2: hipLaunchKernelGGL(mfma_valu_kernel<2>, grid, dim3(256), 0, s, out, iters, 1.f, 1e-3f); break;
    case 4: hipLaunchKernelGGL(mfma_valu_kernel<4>, grid, dim3(256), 0, s, out, iters, 1.f, 1e-3f); break;
    case 6: hipLaunchKernelGGL(mfma_valu_kernel<6>, grid, dim3(256), 0, s, out, iters, 1.f, 1e-3f); break;
    case 8: hipLaunchKernelGGL(mfma_valu_kernel<8>, grid, dim3(256), 0, s, out, iters, 1.f, 1e-3f); break;
    case 12: hipLaunchKernelGGL(mfma_valu_kernel<12>, grid, dim3(256), 0, s, out, iters, 1.f, 1e-3f); break;
    case 16: hipLaunchKernelGGL(mfma_valu_kernel<16>, grid, dim3(256), 0, s, out, iters, 1.f, 1e-3f); break;
    default: return mpsr::fail(MPSR_ERR_INVALID_ARG, "mfma_valu: nv must be 0, 1, 2, 4, 6, 8, 12 or 16");
    }
    MPSR_CHECK_LAUNCH("mfma_valu_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_debug_mfma_lds(float *out, int cus, int waves_per_simd, int mode, int iters, mpsr_stream_t stream)
{
    MPSR_REQUIRE(out && cus > 0 && waves_per_simd >= 1 && waves_per_simd <= 8 && iters > 0, "mfma_lds: bad arguments");
    const dim3 grid((unsigned)(cus * waves_per_simd));
    if (mode == 0) hipLaunchKernelGGL(mfma_lds_kernel<0>, grid, dim3(256), 0, mpsr::as_stream(stream), out, iters);
    else hipLaunchKernelGGL(mfma_lds_kernel<1>, grid, dim3(256), 0, mpsr::as_stream(stream), out, iters);
    MPSR_CHECK_LAUNCH("mfma_lds_kernel");
    return MPSR_OK;
}

// Launches `waves_per_simd` waves on every SIMD of `cus` CUs, each issuing iters * 16 MFMAs.  The caller times it
// (FLOP = cus * 4 * waves_per_simd * iters * 16 * 4096).
extern "C" int mpsr_debug_mfma_peak(float *out, int cus, int waves_per_simd, int chains, int iters, mpsr_stream_t stream)
{
    MPSR_REQUIRE(out && cus > 0 && waves_per_simd >= 1 && waves_per_simd <= 8 && iters > 0 && (chains == 1 || chains == 4),
                 "mfma_peak: bad arguments");
    const dim3 grid((unsigned)(cus * waves_per_simd));
    if (chains == 1) hipLaunchKernelGGL(mfma_peak_kernel<1>, grid, dim3(256), 0, mpsr::as_stream(stream), out, iters, 1.f, 1e-3f);
    else hipLaunchKernelGGL(mfma_peak_kernel<4>, grid, dim3(256), 0, mpsr::as_stream(stream), out, iters, 1.f, 1e-3f);
    MPSR_CHECK_LAUNCH("mfma_peak_kernel");
    return MPSR_OK;
}
