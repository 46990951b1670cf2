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
//     fixed pixel rows for the whole K loop, so a tap is one add + bounds test per row; out-of-image taps and the
//     K tail load zeros.  B tile (BN filters x 32) comes from the (N, K) weight matrix.  16-byte global loads.
//   * Both tiles sit in LDS as [row][32 + 4 pad] floats: the 144-byte row stride makes the ds_read_b128 fragment
//     reads and the ds_write_b128 fills bank-conflict free (MI355X_MICROARCH.md LDS table).
//   * K ordering inside a tile is chosen so a lane's four consecutive floats feed four consecutive MFMAs: lane
//     (i = l & 31, h = l >> 5) reads A[i][8*kb + 4*h .. +3]; MFMA step s multiplies k = 8*kb + 4*h + s.
//   * Software pipeline: next tile's global loads are issued into registers before the current tile's MFMAs and
//     written to LDS after them (one LDS buffer, two barriers per K step); 2-4 workgroups per CU overlap.
//   * Workgroup ids are remapped so consecutive ids of one XCD walk the N tiles of one M panel: the activation
//     panel is fetched from HBM once and re-read from that XCD's L2.
//   * split_k > 1 writes raw partial tiles to a workspace; a second kernel reduces and applies the epilogue
//     (used for the K = 18432 fully-connected layers where M = batch is small).
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 32;
constexpr int LDS_STRIDE = BK + 4;  // floats per LDS row

struct ConvParams {
    const float *x;
    const float *w;
    const float *bias;
    const float *residual;
    float *y;
    float *ws;
    int M, H, W, C, N, KH, KW, dil, relu;
    int ksteps_total, ksteps_per_split, cblocks;
    int mtiles, ntiles, splits;
    unsigned xbytes, wbytes;
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p)
{
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int AV = BM / 32, BV = BN / 32;  // float4 loads per thread per tile
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDS_STRIDE];
    float *As = lds, *Bs = lds + BM * LDS_STRIDE;

    // XCD-aware bijective remap: ids b, b+8, b+16, ... (one XCD) -> consecutive tiles
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int ni = t % p.ntiles;
    const int mi = (t / p.ntiles) % p.mtiles;
    const int si = t / (p.ntiles * p.mtiles);
    const int m0 = mi * BM, n0 = ni * BN;
    const int ks_begin = si * p.ksteps_per_split;
    const int ks_end = min(p.ksteps_total, ks_begin + p.ksteps_per_split);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = tid >> 3, lcol = (tid & 7) * 4;

    // Global loads go through buffer descriptors: a voffset at/after num_records returns zeros in hardware, so an
    // out-of-image tap, a row past M / N and the K tail cost one v_cndmask on the 32-bit offset -- no branch, no
    // select on the data -- and the K loop stays a single basic block.  (Host guarantees both tensors < 4 GiB.)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, (int)p.wbytes, 0x00020000);

    // per-thread A rows: pixel coordinates and byte offset are fixed for the whole K loop
    int ay[AV], ax[AV];
    unsigned abase[AV];
#pragma unroll
    for (int j = 0; j < AV; ++j) {
        const int m = m0 + lrow + 32 * j;
        const bool valid = m < p.M;
        const int pix = valid ? m % (p.H * p.W) : 0;
        ay[j] = valid ? pix / p.W : -(1 << 20);  // an invalid row fails every bounds test
        ax[j] = pix % p.W;
        abase[j] = (unsigned)(valid ? m : 0) * (unsigned)p.C * 4u;
    }
    const int Ktot = p.KH * p.KW * p.C;
    unsigned bbase[BV];
#pragma unroll
    for (int j = 0; j < BV; ++j) {
        const int n = n0 + lrow + 32 * j;
        bbase[j] = n < p.N ? (unsigned)n * (unsigned)Ktot * 4u : p.wbytes;  // past the end -> zeros
    }

    // K-step state, advanced incrementally (channel block fastest, then kx, then ky)
    int st_cb, st_kx, st_ky;
    {
        const int tap = ks_begin / p.cblocks;
        st_cb = ks_begin - tap * p.cblocks;
        st_ky = tap / p.KW;
        st_kx = tap - st_ky * p.KW;
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
            if (++st_kx == p.KW) {
                st_kx = 0;
                ++st_ky;
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < AV; ++j)
            *reinterpret_cast<float4 *>(&As[(lrow + 32 * j) * LDS_STRIDE + lcol]) = ra[j];
#pragma unroll
        for (int j = 0; j < BV; ++j)
            *reinterpret_cast<float4 *>(&Bs[(lrow + 32 * j) * LDS_STRIDE + lcol]) = rb[j];
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

    // K loop (the host guarantees ks_begin < ks_end); last tile peeled so the body has no conditional
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
    if ((p.N & 3) == 0) {
        // Vector path: each wave transposes one 32x32 accumulator tile at a time through its own 4.5 KiB LDS
        // slice so that a lane owns 4 consecutive channels of a pixel: 16-byte residual loads / output stores
        // (4x fewer memory instructions than the element-per-lane layout; the K = 256 conv3 layers are
        // epilogue-bound otherwise).
        __syncthreads();  // every wave is done with the last A/B tile
        float *ep = lds + wave * (32 * LDS_STRIDE);
        const int erow = lane >> 3, ecol = (lane & 7) * 4;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + ecol;
            float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
            if (fused && p.bias && n < p.N) bias = *reinterpret_cast<const float4 *>(p.bias + n);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) ep[((e & 3) + 8 * (e >> 2) + rsub) * LDS_STRIDE + col] = acc[i][j][e];
                __builtin_amdgcn_wave_barrier();
                const int mb = m0 + (wm * TM + i) * 32 + erow;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    float4 v = *reinterpret_cast<const float4 *>(&ep[(erow + 8 * it) * LDS_STRIDE + ecol]);
                    const int m = mb + 8 * it;
                    if (m < p.M && n < p.N) {
                        const size_t o = (size_t)m * p.N + n;
                        if (fused) {
                            v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
                            if (p.residual) {
                                const float4 r = *reinterpret_cast<const float4 *>(p.residual + o);
                                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
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
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + col;
        if (n >= p.N) continue;
        const float bias = (fused && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + (wm * TM + i) * 32 + rsub;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                if (m >= p.M) continue;
                const size_t o = (size_t)m * p.N + n;
                float v = acc[i][j][e];
                if (fused) {
                    v += bias;
                    if (p.residual) v += p.residual[o];
                    if (p.relu) v = fmaxf(v, 0.f);
                }
                dst[o] = v;
            }
        }
    }
}

// y = act(sum_s ws[s] + bias + residual), float4 over n when N % 4 == 0.
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

template <int BM, int BN, int WM, int WN>
int launch(ConvParams &p, hipStream_t s)
{
    p.mtiles = mpsr::ceil_div(p.M, BM);
    p.ntiles = mpsr::ceil_div(p.N, BN);
    const long long blocks = (long long)p.mtiles * p.ntiles * p.splits;
    if (blocks > 0x7fffffffLL) return mpsr::fail(MPSR_ERR_UNSUPPORTED, "conv2d: grid too large");
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    MPSR_CHECK_LAUNCH("conv_igemm_kernel");
    return MPSR_OK;
}

// tuning knob (tests sweep it): -1 = heuristic, 0=128x128 1=128x64 2=64x128 3=64x64 4=128x32
int g_tile_override = -1;
int tile_override() { return g_tile_override; }

}  // namespace

namespace mpsr {

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
    ConvParams p;
    p.x = x; p.w = w; p.bias = bias; p.residual = residual; p.y = y; p.ws = ws;
    p.M = (int)M64; p.H = H; p.W = W; p.C = C; p.N = N; p.KH = KH; p.KW = KW; p.dil = dilation; p.relu = relu;
    p.xbytes = (unsigned)(M64 * C * 4);
    p.wbytes = (unsigned)((long long)N * KH * KW * C * 4);
    p.cblocks = ceil_div(C, BK);
    p.ksteps_total = KH * KW * p.cblocks;
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
    int sel = tile_override();
    if (sel < 0) {
        // measured on MI355X (tools/conv_layer_bench.py, profiles/): 128x128 wins once there are >= ~18 tiles per CU
        // (the 24x24 / 48x48 decoder layers, 130-137 TFLOP/s); the 12x12 trunk layers (M = 36864 at B = 256) only
        // make 2.25 tiles of 128x128 per CU, so 64x64 tiles (9 per CU, 7 waves/SIMD) balance and overlap better
        if (N <= 32) sel = 4;
        else if (p.M < 131072) sel = 3;
        else sel = 0;
    }
    switch (sel) {
        case 0: rc = launch<128, 128, 2, 2>(p, stream); break;
        case 1: rc = launch<128, 64, 2, 2>(p, stream); break;
        case 2: rc = launch<64, 128, 2, 2>(p, stream); break;
        case 3: rc = launch<64, 64, 2, 2>(p, stream); break;
        default: rc = launch<128, 32, 4, 1>(p, stream); break;
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

// Internal (not part of the ABI in monopsr_hip.h): force a tile configuration for tuning and for the tests that
// sweep every instantiation.  Process-wide, not thread-safe.
extern "C" void mpsr_debug_set_conv_tile(int sel) { g_tile_override = sel; }

extern "C" int mpsr_conv2d_nhwc_f32(const float *x, int B, int H, int W, int C, const float *w, const float *bias,
                                    const float *residual, float *y, int N, int KH, int KW, int dilation, int relu,
                                    int split_k, float *ws, size_t ws_floats, mpsr_stream_t stream)
{
    return mpsr::conv2d(x, B, H, W, C, w, bias, residual, y, N, KH, KW, dilation, relu, split_k, ws, ws_floats,
                        mpsr::as_stream(stream));
}
