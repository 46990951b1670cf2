// Backward-pass kernels of the instance path: weight gradient as an fp32-MFMA GEMM reduced
// over pixels, weight re-layout for the data gradient (which then reuses the forward implicit-GEMM kernel),
// bias gradient, ReLU / max-pool / bilinear-resize gradients and a fused Adam step over flat parameter buffers.
// The reference gets all of these from TensorFlow's autodiff (trainer.py:71-81, optimizer_builder.py:61-80); there
// is no reference source to cite beyond the forward call sites.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <cstring>

#include "common.h"

namespace mpsr {
// thin_conv.hip
bool thin_wgrad_applies(int B, int H, int W, int C, int N, int KH, int KW, int dilation);
int thin_wgrad(const float *x, const float *dy, int B, int H, int W, int C, float *dw, float *db, hipStream_t s);
}  // namespace mpsr

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// ------------------------------------------------------------------------------------------------ weight gradient
//
//   dW[n][tap*C + c] += sum_m dY[m][n] * X[pixel(m) + tap][c]
//
// GEMM with the output (N x C per tap) small and the reduction (pixels) long: a workgroup owns a 128 (n) x 128 (c)
// tile of one tap and a contiguous slice of pixels; slices are combined with hardware fp32 atomics into the
// pre-zeroed gradient.  Both operands are read row-wise (a pixel's N / C values are contiguous), staged in LDS as
// [pixel][n] / [pixel][c] and fed to v_mfma_f32_32x32x2_f32 with one ds_read_b32 per operand (lane i reads column
// i of pixel row k: consecutive lanes, consecutive banks).
constexpr int WG_T = 128;  // tile edge (n and c)
constexpr int WG_K = 32;   // pixels per step

struct WgradParams {
    const float *x;
    const float *dy;
    float *dw;
    float *db;  // optional bias gradient: db[n] += sum over pixels of dy[pixel][n]
    int M, H, W, C, N, KH, KW, dil;
    int ntile_n, ntile_c, splits;
    // pixel slices per tap (proportional to the tap's valid pixels) and the cumulative workgroup count; used when
    // the kernel has at most MAX_TAPS taps, otherwise every tap gets `splits` slices
    int per_tap, tap_splits[9], tap_end[9];
    // grouped: workgroup b -> XCD b % 8 (the dispatcher's round-robin); a GROUP = the ntile_n * ntile_c output tiles of
    // one (tap, pixel slice), which all read the same rows of x and dy, is kept on ONE XCD (consecutive workgroups of
    // that XCD), so that the slice comes from HBM once and from that XCD's L2 for the other tiles (r04: -1..3 % per
    // launch: the Infinity Cache already absorbed most of it).  ngroups groups in all, tap_gend[] their cumulative
    // count per tap.
    int grouped, ngroups, tap_gend[9];
    unsigned xbytes, dybytes;
};
constexpr int MAX_TAPS = 9;

#ifndef WG_WAVES
#define WG_WAVES 0  // A/B builds: waves per SIMD the register allocation is held to (0 = hipcc's choice: 136 registers, 3)
#endif
#if WG_WAVES
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WG_WAVES, WG_WAVES))) void conv_wgrad_kernel(const WgradParams p)
#else
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p)
#endif
{
#ifndef WG_DB
#define WG_DB 0  // 1: two LDS tile pairs, one barrier per step.  A/B build (r06), measured SLOWER: 64 KB of LDS leave two
                 // workgroups per CU instead of three -- b3 conv1 232 vs 181 us, the training step 51.2 vs 47.8 ms:
                 // residency, not the second barrier, is what this kernel lives on
#endif
    __shared__ __attribute__((aligned(16))) float At[(WG_DB ? 2 : 1) * WG_K * WG_T];  // [pixel][n]
    __shared__ __attribute__((aligned(16))) float Bt[(WG_DB ? 2 : 1) * WG_K * WG_T];  // [pixel][c]
    int t = blockIdx.x;
    int tap, si, nsplit;
    if (p.grouped) {
        const int tiles = p.ntile_n * p.ntile_c;
        const int xcd = t & 7, l = t >> 3;
        const int g = (l / tiles) * 8 + xcd;
        if (g >= p.ngroups) return;  // block-uniform
        t = l % tiles;
        if (p.per_tap) {
            tap = 0;
            while (g >= p.tap_gend[tap]) ++tap;
            si = g - (tap > 0 ? p.tap_gend[tap - 1] : 0);
            nsplit = p.tap_splits[tap];
        } else {
            const int taps = p.KH * p.KW;
            tap = g % taps;
            si = g / taps;
            nsplit = p.splits;
        }
        t += si * tiles;  // (decoded below: tile column, tile row, slice)
    } else if (p.per_tap) {
        tap = 0;
        while (t >= p.tap_end[tap]) ++tap;
        if (tap > 0) t -= p.tap_end[tap - 1];
        nsplit = p.tap_splits[tap];
    } else {
        tap = t % (p.KH * p.KW);
        t /= p.KH * p.KW;
        nsplit = p.splits;
    }
    const int tn = t % p.ntile_n; t /= p.ntile_n;
    const int tc = t % p.ntile_c; t /= p.ntile_c;
    si = t;
    const int n0 = tn * WG_T, c0 = tc * WG_T;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int dyo = (ky - (p.KH >> 1)) * p.dil, dxo = (kx - (p.KW >> 1)) * p.dil;
    // Only pixels whose shifted partner (y + dyo, x + dxo) lies inside the image contribute to this tap: enumerate
    // that sub-rectangle of every image instead of all pixels (a 3x3 tap at dilation 4 on a 12x12 map touches 44-67 %
    // of them), and split ITS steps over the tap's `nsplit` pixel slices.
    const int ylo = max(0, -dyo), xlo = max(0, -dxo);
    const int hv = p.H - max(0, dyo) - ylo, wv = p.W - max(0, dxo) - xlo;
    if (hv <= 0 || wv <= 0) return;
    const int HWv = hv * wv;
    const float inv_HWv = 1.0f / (float)HWv, inv_wv = 1.0f / (float)wv;
    const int Mv = (p.M / (p.H * p.W)) * HWv;
    const int msteps_tap = (Mv + WG_K - 1) / WG_K;
    const int per_split = (msteps_tap + nsplit - 1) / nsplit;
    const int ms_begin = si * per_split;
    const int ms_end = min(msteps_tap, ms_begin + per_split);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = tid >> 5, lcol = (tid & 31) * 4;  // 8 pixel rows per pass, 32 float4 per row

    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.dy), 0, (int)p.dybytes, 0x00020000);
    const bool nok = n0 + lcol < p.N, cok = c0 + lcol < p.C;
    const int HW = p.H * p.W;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[4], rb[4];
    // Row decode through LDS.  A K step stages 32 pixel rows; which pixel a row is -- (image, y, x) inside the tap's
    // valid rectangle -- and the byte offsets that follow are the same for the 32 threads that share the row.  The 32
    // decodes of a step are done ONCE, by the first half of wave 0, one step ahead, into a two-entry LDS table; every
    // thread then reads its row's pair of offsets and adds its column.  Vector instructions take matrix-pipe time on
    // this hardware (conv_mfma.hip / DESIGN.md 4.1): decoding per thread and per load was ~200 of them per step and
    // wave against 64 MFMAs; this is ~16, plus ~45 on one wave.
    __shared__ uint2 offs[2][WG_K];  // (dy byte offset, x byte offset) of a row, or out-of-range ones past the end
    auto decode_rows = [&](int ms) {
        if (tid < WG_K) {
            const int mv = ms * WG_K + tid;  // index into the tap's valid pixels
            const bool mok = mv < Mv;
            // float reciprocal + one-step fix-up (operands < 2^24, checked on the host; 24-bit multiplies)
            int img = (int)((float)mv * inv_HWv);
            img -= (__mul24(img, HWv) > mv);
            img += (__mul24(img + 1, HWv) <= mv);
            const int r = mv - __mul24(img, HWv);
            int ry = (int)((float)r * inv_wv);
            ry -= (__mul24(ry, wv) > r);
            ry += (__mul24(ry + 1, wv) <= r);
            const int m = __mul24(img, HW) + __mul24(ylo + ry, p.W) + xlo + (r - __mul24(ry, wv));  // the pixel itself
            offs[ms & 1][tid] = make_uint2(
                mok ? (__umul24((unsigned)m, (unsigned)p.N) + (unsigned)n0) * 4u : p.dybytes,
                mok ? (__umul24((unsigned)(m + dyo * p.W + dxo), (unsigned)p.C) + (unsigned)c0) * 4u : p.xbytes);
        }
    };
    const unsigned lcol4 = (unsigned)lcol * 4u;
    auto load_tile = [&](int ms) {
#ifdef WG_NO_LOAD  // (timing experiment only)
        if (ms > ms_begin + 1) return;
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint2 o = offs[ms & 1][lrow + 8 * j];
            // a row past the end keeps an out-of-range offset after the column is added (the host keeps the tensors
            // 4 KiB below 4 GiB)
            const unsigned offa = nok ? o.x + lcol4 : p.dybytes;
            ra[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rdy, offa, 0, 0));
            const unsigned offb = cok ? o.y + lcol4 : p.xbytes;
            rb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, offb, 0, 0));
        }
    };
    auto store_tile = [&](int buf = 0) {
#ifdef WG_NO_LDS_STORE  // (timing experiment only)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(ra[j].x), "v"(ra[j].y), "v"(ra[j].z), "v"(ra[j].w), "v"(rb[j].x), "v"(rb[j].y), "v"(rb[j].z), "v"(rb[j].w));
        return;
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<float4 *>(&At[buf * WG_K * WG_T + (lrow + 8 * j) * WG_T + lcol]) = ra[j];
            *reinterpret_cast<float4 *>(&Bt[buf * WG_K * WG_T + (lrow + 8 * j) * WG_T + lcol]) = rb[j];
        }
    };
    const float *Aw = At + (lane >> 5) * 4 * WG_T + wm * 64 + (lane & 31);
    const float *Bw = Bt + (lane >> 5) * 4 * WG_T + wn * 64 + (lane & 31);
    // bias gradient = column sums of the dy tile, taken by the workgroups of the CENTRE tap / channel tile 0 (they see
    // every pixel exactly once): thread t < 128 owns column n0 + t
    const bool do_db = p.db != nullptr && tap == (p.KH * p.KW) / 2 && tc == 0 && tid < WG_T;
    float colsum = 0.f;
    auto compute_tile = [&](int buf = 0) {
        if (do_db) {
#pragma unroll
            for (int k = 0; k < WG_K; ++k) colsum += At[buf * WG_K * WG_T + k * WG_T + tid];
        }
#pragma unroll
        for (int kb = 0; kb < WG_K / 8; ++kb)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int row = buf * WG_K * WG_T + (kb * 8 + s) * WG_T;
                const float a0 = Aw[row], a1 = Aw[row + 32], b0 = Bw[row], b1 = Bw[row + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
    };

    if (ms_begin >= ms_end) return;  // uniform over the workgroup
    decode_rows(ms_begin);
    decode_rows(ms_begin + 1);
    __syncthreads();
    load_tile(ms_begin);
    store_tile();
    __syncthreads();
#if WG_DB
    int cur = 0;
    for (int ms = ms_begin; ms < ms_end - 1; ++ms) {
        load_tile(ms + 1);
        decode_rows(ms + 2);  // (entry ms & 1: last read by load_tile(ms), before the previous trip's barrier)
        compute_tile(cur);
        store_tile(cur ^ 1);  // the other pair: nobody reads it before the barrier
        __syncthreads();
        cur ^= 1;
    }
    compute_tile(cur);
#else
    for (int ms = ms_begin; ms < ms_end - 1; ++ms) {
        load_tile(ms + 1);
        decode_rows(ms + 2);  // into the table entry load_tile(ms) read before the barriers of the previous trip
        compute_tile();
        __syncthreads();
        store_tile();
        __syncthreads();
    }
    compute_tile();
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[i][j]));

    if (do_db && n0 + tid < p.N) unsafeAtomicAdd(&p.db[n0 + tid], colsum);
    const int Ktot = p.KH * p.KW * p.C;
    const int col = lane & 31, rsub = (lane >> 5) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = c0 + wn * 64 + j * 32 + col;
        if (c >= p.C) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wm * 64 + i * 32 + rsub + (e & 3) + 8 * (e >> 2);
#ifdef WG_PLAIN_STORE  // (timing experiments only, tools/README.md: wrong results)
                if (n < p.N) p.dw[(size_t)n * Ktot + (size_t)tap * p.C + c] = acc[i][j][e];
#elif defined(WG_NO_STORE)
                asm volatile("" ::"v"(acc[i][j][e]));
#else
                if (n < p.N) unsafeAtomicAdd(&p.dw[(size_t)n * Ktot + (size_t)tap * p.C + c], acc[i][j][e]);
#endif
            }
    }
}

// The same GEMM for 1x1 layers WITHOUT shared memory (r06).  Both operands are "K-major" -- a pixel's N / C values are
// contiguous and the reduction runs over pixels -- which is exactly the operand layout of v_mfma_f32_32x32x2_f32: lane
// (i, k) = (lane & 31, lane >> 5) holds column i of reduction row k, i.e. the 32 lanes of a half-wave read ONE 128-byte
// segment of pixel row k.  So a wave feeds its 64 x 64 share of the tile with four dword buffer loads per four MFMAs
// straight from L2 / the vector cache into the MFMA's source registers: no LDS staging, no barriers (the four waves
// of a workgroup never wait for each other), no register -> LDS -> register round trip.  Loads run a ring of WD_DEPTH
// reduction pairs ahead.  Same tiles, slices, XCD grouping, atomics and bias gradient as conv_wgrad_kernel.
// An A/B alternative, not the default (see g_wgrad_direct below): its dword loads cost more than the LDS kernel's
// float4 loads + staging save.
#ifndef WD_DEPTH
#define WD_DEPTH 8
#endif
__global__ __launch_bounds__(256) void pw_wgrad_direct_kernel(const WgradParams p)
{
    int t = blockIdx.x;
    const int nsplit = p.per_tap ? p.tap_splits[0] : p.splits;
    if (p.grouped) {
        const int tiles = p.ntile_n * p.ntile_c;
        const int xcd = t & 7, l = t >> 3;
        const int g = (l / tiles) * 8 + xcd;
        if (g >= p.ngroups) return;  // block-uniform
        t = l % tiles + g * tiles;
    }
    const int tn = t % p.ntile_n; t /= p.ntile_n;
    const int tc = t % p.ntile_c; t /= p.ntile_c;
    const int si = t;
    const int n0 = tn * WG_T, c0 = tc * WG_T;
    const int msteps = (p.M + WG_K - 1) / WG_K;
    const int per_split = (msteps + nsplit - 1) / nsplit;
    const int ms_begin = si * per_split;
    const int ms_end = min(msteps, ms_begin + per_split);
    if (ms_begin >= ms_end) return;
    // reduction pairs (two pixel rows each) of this slice; rows past M read as zero (their offsets are past the buffers)
    const int kp_begin = ms_begin * (WG_K / 2), kp_end = ms_end * (WG_K / 2);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int col = lane & 31, kh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.dy), 0, (int)p.dybytes, 0x00020000);
    const unsigned NB = (unsigned)p.N * 4u, CB = (unsigned)p.C * 4u;
    const int na = n0 + wm * 64 + col, cb = c0 + wn * 64 + col;
    // byte offsets of this lane's element of the first pair; a column outside the tensor keeps an out-of-range offset
    // (increment 0).  The host keeps (M + 2 * WD_DEPTH + 32) rows of either tensor below 4 GiB: no wrap-around.
    const unsigned row0 = (unsigned)(kp_begin * 2 + kh);
    unsigned oa0 = na < p.N ? row0 * NB + (unsigned)na * 4u : 0xfffffff0u;
    unsigned oa1 = na + 32 < p.N ? row0 * NB + (unsigned)(na + 32) * 4u : 0xfffffff0u;
    unsigned ob0 = cb < p.C ? row0 * CB + (unsigned)cb * 4u : 0xfffffff0u;
    unsigned ob1 = cb + 32 < p.C ? row0 * CB + (unsigned)(cb + 32) * 4u : 0xfffffff0u;
    const unsigned ia0 = na < p.N ? 2u * NB : 0u, ia1 = na + 32 < p.N ? 2u * NB : 0u;
    const unsigned ib0 = cb < p.C ? 2u * CB : 0u, ib1 = cb + 32 < p.C ? 2u * CB : 0u;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float a0[WD_DEPTH], a1[WD_DEPTH], b0[WD_DEPTH], b1[WD_DEPTH];
    auto issue = [&](int d) {
        a0[d] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy, oa0, 0, 0));
        a1[d] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy, oa1, 0, 0));
        b0[d] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ob0, 0, 0));
        b1[d] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ob1, 0, 0));
        oa0 += ia0; oa1 += ia1; ob0 += ib0; ob1 += ib1;
    };
    // bias gradient = column sums of dy, by the waves that hold columns n0 .. n0 + 127 of channel tile 0 (wn == 0)
    const bool do_db = p.db != nullptr && tc == 0 && wn == 0;
    float cs0 = 0.f, cs1 = 0.f;
    auto consume = [&](int d) {
        if (do_db) { cs0 += a0[d]; cs1 += a1[d]; }
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[d], b0[d], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[d], b1[d], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[d], b0[d], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[d], b1[d], acc[1][1], 0, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < WD_DEPTH; ++d) issue(d);
    int kp = kp_begin;
    for (; kp + WD_DEPTH <= kp_end; kp += WD_DEPTH) {
#pragma unroll
        for (int d = 0; d < WD_DEPTH; ++d) {
            consume(d);
#ifdef WG_NO_LOAD  // (timing experiment only)
            asm volatile("" : "+v"(a0[d]), "+v"(a1[d]), "+v"(b0[d]), "+v"(b1[d]));
            continue;
#endif
            issue(d);  // the pair WD_DEPTH ahead (past the slice: rows of the next slice or zeros, never consumed)
        }
    }
#pragma unroll
    for (int d = 0; d < WD_DEPTH; ++d)
        if (kp + d < kp_end) consume(d);  // (uniform)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[i][j]));

    if (do_db) {
        cs0 += __shfl_xor(cs0, 32, 64);
        cs1 += __shfl_xor(cs1, 32, 64);
        if (kh == 0) {
            if (na < p.N) unsafeAtomicAdd(&p.db[na], cs0);
            if (na + 32 < p.N) unsafeAtomicAdd(&p.db[na + 32], cs1);
        }
    }
    const int rsub = kh * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = c0 + wn * 64 + j * 32 + col;
        if (c >= p.C) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wm * 64 + i * 32 + rsub + (e & 3) + 8 * (e >> 2);
#ifdef WG_PLAIN_STORE  // (timing experiments only, tools/README.md: wrong results)
                if (n < p.N) p.dw[(size_t)n * p.C + c] = acc[i][j][e];
#elif defined(WG_NO_STORE)
                asm volatile("" ::"v"(acc[i][j][e]));
#else
                if (n < p.N) unsafeAtomicAdd(&p.dw[(size_t)n * p.C + c], acc[i][j][e]);
#endif
            }
    }
}

// wd[c][(T-1-t)*Nd + n] = w[n][t*C + c] (0 for n >= N): the data gradient of a stride-1 SAME convolution is the same
// convolution of dY with the taps flipped and the channel roles swapped.  A transpose per tap: a workgroup moves one
// 64 (n) x 64 (c) tile through LDS, so that the reads run along c and the writes along n (one element per thread
// straight from w[n][..] to wd[c][..] reads with a stride of T C floats: 73 M elements took 0.58 ms of every training
// step).  One launch can serve ONE layer (mpsr_conv2d_dgrad_pack) or every layer of a step (the _batch entry: chunk b
// of the launch belongs to job chunk_job[b]; a job's chunks are numbered (tap, n tile, c tile), c tile fastest).
constexpr int PACK_TILE = 64;
struct PackTableHead {
    long long n_jobs, n_chunks;
};
__device__ __forceinline__ void dgrad_pack_tile(const mpsr_pack_job &j, long long local, float (*tile)[PACK_TILE + 1])
{
    const int nct = (j.C + PACK_TILE - 1) / PACK_TILE, nnt = (j.Nd + PACK_TILE - 1) / PACK_TILE;
    const int ct = (int)(local % nct);
    const long long r = local / nct;
    const int nt = (int)(r % nnt), t = (int)(r / nnt);
    const int c0 = ct * PACK_TILE, n0 = nt * PACK_TILE;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
#pragma unroll 4
    for (int k = 0; k < PACK_TILE / 4; ++k) {
        const int n = n0 + ly + 4 * k, c = c0 + lx;
        tile[ly + 4 * k][lx] = (n < j.N && c < j.C) ? j.w[((size_t)n * j.T + t) * j.C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < PACK_TILE / 4; ++k) {
        const int c = c0 + ly + 4 * k, n = n0 + lx;
        if (c < j.C && n < j.Nd) j.wd[((size_t)c * j.T + (j.T - 1 - t)) * j.Nd + n] = tile[lx][ly + 4 * k];
    }
}

__global__ __launch_bounds__(256) void dgrad_pack_kernel(const mpsr_pack_job j)
{
    __shared__ float tile[PACK_TILE][PACK_TILE + 1];
    dgrad_pack_tile(j, blockIdx.x, tile);
}

__global__ __launch_bounds__(256) void dgrad_pack_batch_kernel(const char *__restrict__ table)
{
    __shared__ float tile[PACK_TILE][PACK_TILE + 1];
    const PackTableHead *head = reinterpret_cast<const PackTableHead *>(table);
    const mpsr_pack_job *jobs = reinterpret_cast<const mpsr_pack_job *>(table + sizeof(PackTableHead));
    const int *chunk_job = reinterpret_cast<const int *>(table + sizeof(PackTableHead) + head->n_jobs * sizeof(mpsr_pack_job));
    const mpsr_pack_job j = jobs[chunk_job[blockIdx.x]];
    dgrad_pack_tile(j, (long long)blockIdx.x - j.chunk0, tile);
}

// ReLU mask of a tensor as bits: bit b of bits[(m >> 5) * N + n] = (y[m][n] > 0) for m = 32 (m >> 5) + mask_row(b);
// rows past M read as zero.  mask_row(b) = (b & 3) + 8 ((b >> 2) & 3) + 4 (b >> 4) is the accumulator layout of the
// 32x32 MFMA (a lane's 16 elements of a 32-row block are one halfword), so that the pointwise kernel (pointwise.hip) can
// write the same words from its forward epilogue and apply them in the store path of a data-gradient launch: the
// gradient then leaves already multiplied by the mask of the tensor it belongs to.  A thread owns 32 rows x 4 columns.
__device__ __forceinline__ int mask_row(int b) { return (b & 3) + 8 * ((b >> 2) & 3) + 4 * (b >> 4); }

__global__ __launch_bounds__(256) void relu_bitmask_kernel(const float *__restrict__ y, long long M, int N,
                                                           unsigned *__restrict__ bits, long long total)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int groups = N >> 2;
    const long long g = t / groups;
    const int cg = (int)(t - g * groups);
    const float4 *src = reinterpret_cast<const float4 *>(y) + (size_t)g * 32 * groups + cg;
    const int rows = (int)min(32LL, M - g * 32);
    uint4 w = make_uint4(0u, 0u, 0u, 0u);
    if (rows == 32) {
        float4 v[32];
#pragma unroll
        for (int b = 0; b < 32; ++b) v[b] = src[(size_t)mask_row(b) * groups];
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            w.x |= (v[b].x > 0.f ? 1u : 0u) << b;
            w.y |= (v[b].y > 0.f ? 1u : 0u) << b;
            w.z |= (v[b].z > 0.f ? 1u : 0u) << b;
            w.w |= (v[b].w > 0.f ? 1u : 0u) << b;
        }
    } else {
        for (int b = 0; b < 32; ++b) {
            if (mask_row(b) >= rows) continue;
            const float4 v = src[(size_t)mask_row(b) * groups];
            w.x |= (v.x > 0.f ? 1u : 0u) << b;
            w.y |= (v.y > 0.f ? 1u : 0u) << b;
            w.z |= (v.z > 0.f ? 1u : 0u) << b;
            w.w |= (v.w > 0.f ? 1u : 0u) << b;
        }
    }
    reinterpret_cast<uint4 *>(bits)[(size_t)g * groups + cg] = w;
}

// Fused activation + bias gradient: g = (y > 0 ? dy : 0) (or g = dy when y == nullptr), written to dx (may be
// nullptr when only the column sums are wanted), and db[n] += sum_m g[m][n].  One pass over dy: a thread owns one
// float4 column group and strides over rows; column sums are combined across the workgroup in LDS, then one fp32
// atomic per column per workgroup.  N % 4 == 0, N <= 1024.
__global__ __launch_bounds__(256) void act_bias_grad_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                            float *__restrict__ dx, float *__restrict__ db,
                                                            long long M, int N, long long rows_per_block)
{
    __shared__ float4 part[256];
    const int groups = N >> 2;            // float4 column groups per row
    const int rstride = 256 / groups;     // rows covered per pass (>= 1 since N <= 1024)
    const int tid = threadIdx.x;
    const int cg = tid % groups, r0 = tid / groups;
    const long long m0 = (long long)blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 < rstride) {
        for (long long m = m0 + r0; m < m1; m += rstride) {
            const size_t o = (size_t)m * groups + cg;
            float4 g = reinterpret_cast<const float4 *>(dy)[o];
            if (y) {
                const float4 a = reinterpret_cast<const float4 *>(y)[o];
                g.x = a.x > 0.f ? g.x : 0.f;
                g.y = a.y > 0.f ? g.y : 0.f;
                g.z = a.z > 0.f ? g.z : 0.f;
                g.w = a.w > 0.f ? g.w : 0.f;
            }
            if (dx) reinterpret_cast<float4 *>(dx)[o] = g;
            s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
        }
    }
    if (!db) return;
    part[tid] = s;
    __syncthreads();
    if (tid < groups) {
        float4 t = part[tid];
        for (int r = 1; r < rstride; ++r) {
            const float4 u = part[r * groups + tid];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        unsafeAtomicAdd(&db[4 * tid], t.x);
        unsafeAtomicAdd(&db[4 * tid + 1], t.y);
        unsafeAtomicAdd(&db[4 * tid + 2], t.z);
        unsafeAtomicAdd(&db[4 * tid + 3], t.w);
    }
}

// generic fallbacks (any N)
__global__ __launch_bounds__(256) void bias_grad_kernel(const float *__restrict__ dy, long long M, int N,
                                                        long long rows_per_block, float *__restrict__ db)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const long long m0 = (long long)blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
    float s = 0.f;
    for (long long m = m0; m < m1; ++m) s += dy[m * N + n];
    unsafeAtomicAdd(&db[n], s);
}

__global__ __launch_bounds__(256) void relu_grad_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                        float *__restrict__ dx, long long total)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x)
        dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// max-pool gradient: each output cell routes its gradient to the first maximum of its window (row-major scan)
__global__ __launch_bounds__(256) void max_pool_grad_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                            int H, int W, int C, int OH, int OW, int k, int s,
                                                            int pad_top, int pad_left, float *__restrict__ dx,
                                                            long long total)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const int b = (int)(r / OH);
        const float *xb = x + (size_t)b * H * W * C;
        float best = -__builtin_inff();
        int by = -1, bx = -1;
        for (int ky = 0; ky < k; ++ky) {
            const int y = oy * s - pad_top + ky;
            if (y < 0 || y >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int xx = ox * s - pad_left + kx;
                if (xx < 0 || xx >= W) continue;
                const float v = xb[((size_t)y * W + xx) * C + c];
                if (v > best) {
                    best = v;
                    by = y;
                    bx = xx;
                }
            }
        }
        if (by >= 0) unsafeAtomicAdd(&dx[(((size_t)b * H + by) * W + bx) * C + c], dy[i]);
    }
}

// bilinear-resize gradient (TF-1.8 kernel geometry) in GATHER form: a thread owns one float4 of dx and sums the output
// gradients that read it.  Along each axis output o reads inputs i0 = floor(o*scale) (weight 1-l) and
// i1 = min(i0+1, I-1) (weight l); the candidates for input i are the o with o*scale in (i-1, i+1), found by scanning
// a slightly wider window and re-evaluating the forward's own float expressions (so the weights match it exactly).
// No atomics: deterministic, and dx needs no pre-zeroing.
constexpr int RG_MAX = 12;  // candidates kept per axis; the host falls back to the scatter kernel beyond that

__device__ __forceinline__ int resize_axis_taps(int i, int I, int O, float scale, int *os, float *ws)
{
    const float inv = 1.0f / scale;
    int lo = (int)floorf((float)(i - 1) * inv) - 1, hi = (int)ceilf((float)(i + 1) * inv) + 1;
    lo = max(lo, 0);
    hi = min(hi, O - 1);
    int n = 0;
    for (int o = lo; o <= hi; ++o) {
        const float sp = (float)o * scale;
        const int i0 = (int)floorf(sp);
        const int i1 = min(i0 + 1, I - 1);
        const float l = sp - (float)i0;
        const float w = (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
        if (w != 0.f && n < RG_MAX) {
            os[n] = o;
            ws[n] = w;
            ++n;
        }
    }
    return n;
}

__global__ __launch_bounds__(256) void resize_bilinear_grad_gather_kernel(const float *__restrict__ dy, int H, int W,
                                                                          int C4, int OH, int OW, float hscale,
                                                                          float wscale, float *__restrict__ dx,
                                                                          long long total)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int ix = (int)(r % W);
        r /= W;
        const int iy = (int)(r % H);
        const int b = (int)(r / H);
        int oys[RG_MAX], oxs[RG_MAX];
        float wys[RG_MAX], wxs[RG_MAX];
        const int ny = resize_axis_taps(iy, H, OH, hscale, oys, wys);
        const int nx = resize_axis_taps(ix, W, OW, wscale, oxs, wxs);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 *src = reinterpret_cast<const float4 *>(dy) + (size_t)b * OH * OW * C4 + c;
        for (int a = 0; a < ny; ++a)
            for (int e = 0; e < nx; ++e) {
                const float4 g = src[((size_t)oys[a] * OW + oxs[e]) * C4];
                const float w = wys[a] * wxs[e];
                acc.x += w * g.x;
                acc.y += w * g.y;
                acc.z += w * g.z;
                acc.w += w * g.w;
            }
        reinterpret_cast<float4 *>(dx)[i] = acc;
    }
}

// scatter form (any C, any scale): each output gradient is added to its 4 source pixels; dx pre-zeroed
__global__ __launch_bounds__(256) void resize_bilinear_grad_kernel(const float *__restrict__ dy, int H, int W, int C,
                                                                   int OH, int OW, float hscale, float wscale,
                                                                   float *__restrict__ dx, long long total)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int ox = (int)(r % OW);
        r /= OW;
        const int oy = (int)(r % OH);
        const int b = (int)(r / OH);
        const float sy = (float)oy * hscale, sx = (float)ox * wscale;
        const int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float yl = sy - (float)y0, xl = sx - (float)x0;
        const float g = dy[i];
        float *base = dx + (size_t)b * H * W * C + c;
        unsafeAtomicAdd(base + ((size_t)y0 * W + x0) * C, g * (1.f - yl) * (1.f - xl));
        unsafeAtomicAdd(base + ((size_t)y0 * W + x1) * C, g * (1.f - yl) * xl);
        unsafeAtomicAdd(base + ((size_t)y1 * W + x0) * C, g * yl * (1.f - xl));
        unsafeAtomicAdd(base + ((size_t)y1 * W + x1) * C, g * yl * xl);
    }
}

// Adam (tf.train.AdamOptimizer form: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps)) over flat buffers
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, long long n,
                                                   float lr_t, float b1, float b2, float eps, float grad_scale,
                                                   const float *__restrict__ lr_t_dev)
{
    if (lr_t_dev) lr_t = *lr_t_dev;  // (a captured launch reads the step's rate from memory: mpsr_adam_step_lr_dev)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

int g_wgrad_grouped = 1;  // mpsr_debug_set_wgrad_grouped: the XCD-aware workgroup order (A/B)
// mpsr_debug_set_wgrad_direct: 1x1 layers on pw_wgrad_direct_kernel.  Measured and OFF (r06, tools/wgrad_bench.py [--lds],
// profiles/r06_wgrad_ab.txt): b3 conv1 198 vs 188 us, conv3 190 vs 181 -- the ring's depth (4 / 8 / 16 pairs) changes
// nothing, and with loads AND stores knocked out both kernels sit at 158-165 us: the loop is matrix-bound at the
// clock the board sustains, what a launch pays on top is its fixed ramp-up / drain (~30 us), not LDS staging.
int g_wgrad_direct = 0;
int g_wgrad_min_steps = 12;  // mpsr_debug_set_wgrad_min_steps: 32-pixel steps a slice of a 1x1 layer reduces at least (0: off)
int g_wgrad_target = 768;  // one round of the 3 workgroups per CU the kernel's registers allow (measured best)

inline int grid_for(long long total) { return (int)((total + 255) / 256 < 262144 ? (total + 255) / 256 : 262144); }

}  // namespace

extern "C" int mpsr_conv2d_wgrad_f32(const float *x, const float *dy, int B, int H, int W, int C, int N, int KH, int KW,
                                     int dilation, float *dw, float *db, mpsr_stream_t stream)
{
    MPSR_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && N > 0 && (KH & 1) && (KW & 1) && dilation >= 1,
                 "conv2d_wgrad: bad shape");
    MPSR_REQUIRE(C % 4 == 0 && N % 4 == 0, "conv2d_wgrad: C=%d and N=%d must be multiples of 4", C, N);
    if (B == 0) return MPSR_OK;
    MPSR_REQUIRE(x && dy && dw, "conv2d_wgrad: null pointer");
    // a head with (at most) four output channels: reduce over the pixels on the vector ALU (thin_conv.hip)
    // (it reads dy as 16-byte pixels)
    if (((uintptr_t)dy & 15) == 0 && mpsr::thin_wgrad_applies(B, H, W, C, N, KH, KW, dilation))
        return mpsr::thin_wgrad(x, dy, B, H, W, C, dw, db, mpsr::as_stream(stream));
    const long long M = (long long)B * H * W;
    MPSR_REQUIRE(M * C * 4 < 0xfffff000LL && M * N * 4 < 0xfffff000LL, "conv2d_wgrad: tensor exceeds 4 GiB");
    // the kernel decodes pixel indices with 24-bit multiplies and a float reciprocal
    MPSR_REQUIRE(M < (1LL << 24) && N < (1 << 24) && C < (1 << 24),
                 "conv2d_wgrad: more than 2^24 pixels (or channels) in one call; split the batch");
    WgradParams p;
    p.x = x; p.dy = dy; p.dw = dw; p.db = db;
    p.M = (int)M; p.H = H; p.W = W; p.C = C; p.N = N; p.KH = KH; p.KW = KW; p.dil = dilation;
    p.xbytes = (unsigned)(M * C * 4);
    p.dybytes = (unsigned)(M * N * 4);
    p.ntile_n = mpsr::ceil_div(N, WG_T);
    p.ntile_c = mpsr::ceil_div(C, WG_T);
    const int taps = KH * KW;
    const int tiles = p.ntile_n * p.ntile_c;
    const int msteps_total = mpsr::ceil_div(p.M, WG_K);
    int slices = g_wgrad_target / (tiles * taps > 0 ? tiles * taps : 1);  // aim at one round of resident workgroups
    if (slices < 1) slices = 1;
    // XCD-aware order (a group = the tiles of one pixel slice, all on one XCD): whole groups must fill an XCD's share of
    // the round.  18 tiles x 42 slices put 6 groups = 108 workgroups on two of the XCDs' 96 slots -- a second round for
    // 12 of them: the decoder's tap-GEMM weight gradients ran at 73 / 93 TFLOP/s where the same kernel without the order
    // reaches 118 / 119.  So: as many whole groups per XCD as fit, and with fewer than two the plain order.
    bool group_ok = true;
    if (taps == 1 && tiles > 1) {
        const int per_xcd = (g_wgrad_target / 8) / tiles;
        if (per_xcd >= 2) slices = 8 * per_xcd;
        else group_ok = false;
    }
    // Short reductions (r06, tools/wgrad_small_m.py / wgrad_bench.py --target): a slice pays its 128 x 128 atomics whatever
    // it reduced, and the slices of a tile contend for the same addresses -- below ~12 steps of 32 pixels per slice more
    // slices cost more than they parallelise.  One 40x152 map (190 steps): 61.6 us at 48 slices, 44.9 at 16; block2's
    // conv3 at 256 crops (4 tiles: 192 slices of 6 steps): 84.4 -> 67.6 us at 64.  Floor: 128 workgroups in all (a launch
    // that small is latency-bound and wants the parallelism back: 128 -> 512 on one map 25.5 us at 32 slices, 28.4 at 24).
    if (taps == 1 && g_wgrad_min_steps > 0) {
        int cap = msteps_total / g_wgrad_min_steps;
        const int floor_slices = mpsr::ceil_div(128, tiles);
        if (cap < floor_slices) cap = floor_slices;
        if (tiles > 1 && group_ok) cap = (cap + 7) / 8 * 8;  // whole groups per XCD
        if (slices > cap) slices = cap;
    }
    if (slices > msteps_total) slices = msteps_total;
    p.splits = slices;
    p.per_tap = taps <= MAX_TAPS;
    long long blocks = (long long)tiles * taps * slices;
    long long groups = (long long)taps * slices;
    if (p.per_tap) {
        // share taps * slices pixel slices among the taps in proportion to their valid pixels
        long long valid[MAX_TAPS], sum = 0;
        for (int t = 0; t < taps; ++t) {
            const int dyo = (t / KW - (KH >> 1)) * dilation, dxo = (t % KW - (KW >> 1)) * dilation;
            const long long hv = H - (dyo < 0 ? -dyo : dyo), wv = W - (dxo < 0 ? -dxo : dxo);
            valid[t] = hv > 0 && wv > 0 ? hv * wv : 0;
            sum += valid[t];
        }
        blocks = 0;
        for (int t = 0; t < taps; ++t) {
            long long st = sum > 0 ? (valid[t] * taps * slices + sum / 2) / sum : 0;
            const long long steps = (valid[t] * B + WG_K - 1) / WG_K;
            if (st > steps) st = steps;
            if (st < 1) st = 1;  // an empty tap exits at once
            p.tap_splits[t] = (int)st;
            blocks += st * tiles;
            p.tap_end[t] = (int)blocks;
            p.tap_gend[t] = (int)(blocks / tiles);
        }
        groups = blocks / tiles;
    }
    p.grouped = g_wgrad_grouped && group_ok && groups >= 8 && tiles > 1;
    p.ngroups = (int)groups;
    if (p.grouped) blocks = (groups + 7) / 8 * 8 * tiles;
    // 1x1 layers: operands straight from memory into the MFMA's source registers (no wrap-around of its running
    // 32-bit offsets: the ring reads up to 2 * WD_DEPTH rows past a slice, a slice ends up to 31 rows past M)
    const long long slack = 2 * WD_DEPTH + WG_K;
    if (g_wgrad_direct && taps == 1 && (M + slack) * C * 4 < 0xfffffff0LL && (M + slack) * N * 4 < 0xfffffff0LL) {
        hipLaunchKernelGGL(pw_wgrad_direct_kernel, dim3((unsigned)blocks), dim3(256), 0, mpsr::as_stream(stream), p);
        MPSR_CHECK_LAUNCH("pw_wgrad_direct_kernel");
        return MPSR_OK;
    }
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, mpsr::as_stream(stream), p);
    MPSR_CHECK_LAUNCH("conv_wgrad_kernel");
    return MPSR_OK;
}

static long long pack_chunks(const mpsr_pack_job &j)
{
    return (long long)j.T * ((j.Nd + PACK_TILE - 1) / PACK_TILE) * ((j.C + PACK_TILE - 1) / PACK_TILE);
}

extern "C" int mpsr_conv2d_dgrad_pack(const float *w, int N, int KH, int KW, int C, float *wd, mpsr_stream_t stream)
{
    MPSR_REQUIRE(N > 0 && KH > 0 && KW > 0 && C > 0 && w && wd, "conv2d_dgrad_pack: bad argument");
    const mpsr_pack_job j{w, wd, N, N, KH * KW, C, 0};
    const long long chunks = pack_chunks(j);
    MPSR_REQUIRE(chunks < (1ll << 31), "conv2d_dgrad_pack: filter too large for one launch");
    hipLaunchKernelGGL(dgrad_pack_kernel, dim3((unsigned)chunks), dim3(256), 0, mpsr::as_stream(stream), j);
    MPSR_CHECK_LAUNCH("dgrad_pack_kernel");
    return MPSR_OK;
}

static bool pack_jobs_ok(const mpsr_pack_job *jobs, int n_jobs)
{
    if (!jobs || n_jobs <= 0) return false;
    for (int i = 0; i < n_jobs; ++i)
        if (!jobs[i].w || !jobs[i].wd || jobs[i].N <= 0 || jobs[i].Nd < jobs[i].N || jobs[i].T <= 0 || jobs[i].C <= 0) return false;
    return true;
}

extern "C" size_t mpsr_dgrad_pack_table_bytes(const mpsr_pack_job *jobs, int n_jobs)
{
    if (!pack_jobs_ok(jobs, n_jobs)) return 0;
    long long chunks = 0;
    for (int i = 0; i < n_jobs; ++i) chunks += pack_chunks(jobs[i]);
    return sizeof(PackTableHead) + (size_t)n_jobs * sizeof(mpsr_pack_job) + (size_t)chunks * sizeof(int);
}

extern "C" int mpsr_dgrad_pack_table_build(const mpsr_pack_job *jobs, int n_jobs, void *table_host, long long *n_chunks)
{
    MPSR_REQUIRE(pack_jobs_ok(jobs, n_jobs) && table_host && n_chunks, "dgrad_pack_table_build: bad argument");
    char *t = static_cast<char *>(table_host);
    mpsr_pack_job *out = reinterpret_cast<mpsr_pack_job *>(t + sizeof(PackTableHead));
    int *chunk_job = reinterpret_cast<int *>(t + sizeof(PackTableHead) + (size_t)n_jobs * sizeof(mpsr_pack_job));
    long long chunk = 0;
    for (int i = 0; i < n_jobs; ++i) {
        out[i] = jobs[i];
        out[i].chunk0 = chunk;
        for (long long k = pack_chunks(jobs[i]); k > 0; --k) chunk_job[chunk++] = i;
    }
    MPSR_REQUIRE(chunk < (1ll << 31), "dgrad_pack_table_build: too many chunks for one launch");
    PackTableHead head{n_jobs, chunk};
    memcpy(t, &head, sizeof(head));
    *n_chunks = chunk;
    return MPSR_OK;
}

extern "C" int mpsr_conv2d_dgrad_pack_batch(const void *table_dev, long long n_chunks, mpsr_stream_t stream)
{
    MPSR_REQUIRE(table_dev && n_chunks > 0 && n_chunks < (1ll << 31), "conv2d_dgrad_pack_batch: bad argument");
    hipLaunchKernelGGL(dgrad_pack_batch_kernel, dim3((unsigned)n_chunks), dim3(256), 0, mpsr::as_stream(stream),
                       static_cast<const char *>(table_dev));
    MPSR_CHECK_LAUNCH("dgrad_pack_batch_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_bias_grad(const float *dy, long long M, int N, float *db, mpsr_stream_t stream)
{
    MPSR_REQUIRE(M >= 0 && N > 0, "bias_grad: bad shape");
    if (M == 0) return MPSR_OK;
    MPSR_REQUIRE(dy && db, "bias_grad: null pointer");
    long long splits = (M + 511) / 512;
    if (splits > 2048) splits = 2048;
    const long long rows = (M + splits - 1) / splits;
    hipLaunchKernelGGL(bias_grad_kernel, dim3(mpsr::ceil_div(N, 256), (unsigned)((M + rows - 1) / rows)), dim3(256), 0,
                       mpsr::as_stream(stream), dy, M, N, rows, db);
    MPSR_CHECK_LAUNCH("bias_grad_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_act_bias_grad(const float *dy, const float *y, float *dx, float *db, long long M, int N,
                                  mpsr_stream_t stream)
{
    MPSR_REQUIRE(M >= 0 && N > 0, "act_bias_grad: bad shape");
    if (M == 0) return MPSR_OK;
    MPSR_REQUIRE(dy && (dx || db), "act_bias_grad: null pointer");
    hipStream_t s = mpsr::as_stream(stream);
    if (N % 4 == 0 && N <= 1024 && 256 % (N / 4) == 0 && ((uintptr_t)dy & 15) == 0 && (!y || ((uintptr_t)y & 15) == 0) &&
        (!dx || ((uintptr_t)dx & 15) == 0)) {
        long long blocks = (M * N / 4 + 256 * 16 - 1) / (256 * 16);  // ~16 float4 per thread
        if (blocks < 1) blocks = 1;
        if (blocks > 4096) blocks = 4096;
        const long long rows = (M + blocks - 1) / blocks;
        hipLaunchKernelGGL(act_bias_grad_kernel, dim3((unsigned)((M + rows - 1) / rows)), dim3(256), 0, s, dy, y, dx, db,
                           M, N, rows);
        MPSR_CHECK_LAUNCH("act_bias_grad_kernel");
        return MPSR_OK;
    }
    const float *g = dy;
    if (y) {
        MPSR_REQUIRE(dx, "act_bias_grad: dx required on the generic path when y is given");
        hipLaunchKernelGGL(relu_grad_kernel, dim3(grid_for(M * N)), dim3(256), 0, s, dy, y, dx, M * N);
        MPSR_CHECK_LAUNCH("relu_grad_kernel");
        g = dx;
    }
    if (db) {
        long long splits = (M + 511) / 512;
        if (splits > 2048) splits = 2048;
        const long long rows = (M + splits - 1) / splits;
        hipLaunchKernelGGL(bias_grad_kernel, dim3(mpsr::ceil_div(N, 256), (unsigned)((M + rows - 1) / rows)), dim3(256), 0,
                           s, g, M, N, rows, db);
        MPSR_CHECK_LAUNCH("bias_grad_kernel");
    }
    return MPSR_OK;
}

namespace mpsr {
bool pointwise_masked_applies(long long M, int K, int N);
int conv1x1_pointwise_masked(const float *x, long long M, int K, const float *w, const float *bias, const float *residual,
                             const unsigned *mask, float *y, int N, hipStream_t s);
int conv1x1_pointwise_emit(const float *x, long long M, int K, const float *w, const float *bias, const float *residual,
                           int relu, float *y, unsigned *bits, int N, hipStream_t s);
}  // namespace mpsr

extern "C" long long mpsr_relu_bitmask_words(long long M, int N) { return M > 0 && N > 0 ? (M + 31) / 32 * N : 0; }

extern "C" int mpsr_relu_bitmask(const float *y, long long M, int N, unsigned *bits, mpsr_stream_t stream)
{
    MPSR_REQUIRE(M >= 0 && N > 0 && N % 4 == 0, "relu_bitmask: N=%d must be a positive multiple of 4", N);
    if (M == 0) return MPSR_OK;
    MPSR_REQUIRE(y && bits && ((uintptr_t)y & 15) == 0 && ((uintptr_t)bits & 15) == 0, "relu_bitmask: null or unaligned pointer");
    const long long total = (M + 31) / 32 * (N / 4);
    MPSR_REQUIRE((total + 255) / 256 < 0x7fffffffLL, "relu_bitmask: tensor too large");
    hipLaunchKernelGGL(relu_bitmask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, mpsr::as_stream(stream), y,
                       M, N, bits, total);
    MPSR_CHECK_LAUNCH("relu_bitmask_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_conv1x1_masked_applies(long long M, int K, int N)
{
    return mpsr_get_conv_math() == 0 && mpsr::pointwise_masked_applies(M, K, N) ? 1 : 0;
}

extern "C" int mpsr_conv1x1_masked_f32(const float *x, long long M, int K, const float *w, const float *bias,
                                       const float *residual, const unsigned *mask, float *y, int N, mpsr_stream_t stream)
{
    MPSR_REQUIRE(M >= 0 && K > 0 && N > 0, "conv1x1_masked: bad shape");
    if (M == 0) return MPSR_OK;
    MPSR_REQUIRE(x && w && mask && y, "conv1x1_masked: null pointer");
    if (!mpsr_conv1x1_masked_applies(M, K, N))
        return mpsr::fail(MPSR_ERR_UNSUPPORTED, "conv1x1_masked: shape M=%lld K=%d N=%d (or the arithmetic mode) is not taken; "
                                                "ask mpsr_conv1x1_masked_applies first", M, K, N);
    return mpsr::conv1x1_pointwise_masked(x, M, K, w, bias, residual, mask, y, N, mpsr::as_stream(stream));
}

extern "C" int mpsr_conv1x1_relu_bitmask_f32(const float *x, long long M, int K, const float *w, const float *bias,
                                             const float *residual, int relu, float *y, unsigned *bits, int N,
                                             mpsr_stream_t stream)
{
    MPSR_REQUIRE(M >= 0 && K > 0 && N > 0, "conv1x1_relu_bitmask: bad shape");
    if (M == 0) return MPSR_OK;
    MPSR_REQUIRE(x && w && bits && y, "conv1x1_relu_bitmask: null pointer");
    if (!mpsr_conv1x1_masked_applies(M, K, N))
        return mpsr::fail(MPSR_ERR_UNSUPPORTED, "conv1x1_relu_bitmask: shape M=%lld K=%d N=%d (or the arithmetic mode) is not "
                                                "taken; ask mpsr_conv1x1_masked_applies first", M, K, N);
    return mpsr::conv1x1_pointwise_emit(x, M, K, w, bias, residual, relu, y, bits, N, mpsr::as_stream(stream));
}

extern "C" int mpsr_relu_grad(const float *dy, const float *y, float *dx, long long total, mpsr_stream_t stream)
{
    MPSR_REQUIRE(total >= 0, "relu_grad: bad size");
    if (total == 0) return MPSR_OK;
    MPSR_REQUIRE(dy && y && dx, "relu_grad: null pointer");
    hipLaunchKernelGGL(relu_grad_kernel, dim3(grid_for(total)), dim3(256), 0, mpsr::as_stream(stream), dy, y, dx, total);
    MPSR_CHECK_LAUNCH("relu_grad_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_max_pool_grad(const float *x, const float *dy, int B, int H, int W, int C, int k, int s_,
                                  int pad_same, float *dx, mpsr_stream_t stream)
{
    MPSR_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && k > 0 && s_ > 0, "max_pool_grad: bad shape");
    if (B == 0) return MPSR_OK;
    MPSR_REQUIRE(x && dy && dx, "max_pool_grad: null pointer");
    int OH, OW, pt = 0, pl = 0;
    if (pad_same) {
        OH = mpsr::ceil_div(H, s_);
        OW = mpsr::ceil_div(W, s_);
        const int th = (OH - 1) * s_ + k - H, tw = (OW - 1) * s_ + k - W;
        pt = (th > 0 ? th : 0) / 2;
        pl = (tw > 0 ? tw : 0) / 2;
    } else {
        OH = (H - k) / s_ + 1;
        OW = (W - k) / s_ + 1;
    }
    hipStream_t s = mpsr::as_stream(stream);
    MPSR_CHECK_HIP(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)B * H * W * C, s));
    const long long total = (long long)B * OH * OW * C;
    hipLaunchKernelGGL(max_pool_grad_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, dy, H, W, C, OH, OW, k, s_, pt,
                       pl, dx, total);
    MPSR_CHECK_LAUNCH("max_pool_grad_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_resize_bilinear_grad(const float *dy, int B, int H, int W, int C, int OH, int OW, int align_corners,
                                         float *dx, mpsr_stream_t stream)
{
    MPSR_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0, "resize_bilinear_grad: bad shape");
    if (B == 0) return MPSR_OK;
    MPSR_REQUIRE(dy && dx, "resize_bilinear_grad: null pointer");
    const float hscale = (align_corners && OH > 1) ? (float)(H - 1) / (float)(OH - 1) : (float)H / (float)OH;
    const float wscale = (align_corners && OW > 1) ? (float)(W - 1) / (float)(OW - 1) : (float)W / (float)OW;
    hipStream_t s = mpsr::as_stream(stream);
    // gather form when rows are float4-addressable and at most RG_MAX outputs read one input along an axis
    const bool few = 2.0f / hscale + 4.0f <= (float)RG_MAX && 2.0f / wscale + 4.0f <= (float)RG_MAX;
    if (few && C % 4 == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0) {
        const long long n4 = (long long)B * H * W * (C / 4);
        hipLaunchKernelGGL(resize_bilinear_grad_gather_kernel, dim3(grid_for(n4)), dim3(256), 0, s, dy, H, W, C / 4, OH,
                           OW, hscale, wscale, dx, n4);
        MPSR_CHECK_LAUNCH("resize_bilinear_grad_gather_kernel");
        return MPSR_OK;
    }
    MPSR_CHECK_HIP(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)B * H * W * C, s));
    const long long total = (long long)B * OH * OW * C;
    hipLaunchKernelGGL(resize_bilinear_grad_kernel, dim3(grid_for(total)), dim3(256), 0, s, dy, H, W, C, OH, OW, hscale,
                       wscale, dx, total);
    MPSR_CHECK_LAUNCH("resize_bilinear_grad_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_adam_step(float *param, const float *grad, float *m, float *v, long long n, float lr, float beta1,
                              float beta2, float eps, int step, float grad_scale, mpsr_stream_t stream)
{
    MPSR_REQUIRE(n >= 0 && step >= 1, "adam_step: bad argument");
    if (n == 0) return MPSR_OK;
    MPSR_REQUIRE(param && grad && m && v, "adam_step: null pointer");
    const double c1 = 1.0 - pow((double)beta1, step), c2 = 1.0 - pow((double)beta2, step);
    const float lr_t = (float)(lr * sqrt(c2) / c1);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, mpsr::as_stream(stream), param, grad, m, v, n, lr_t,
                       beta1, beta2, eps, grad_scale, (const float *)nullptr);
    MPSR_CHECK_LAUNCH("adam_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_adam_step_lr_dev(float *param, const float *grad, float *m, float *v, long long n,
                                     const float *lr_t_dev, float beta1, float beta2, float eps, float grad_scale,
                                     mpsr_stream_t stream)
{
    MPSR_REQUIRE(n >= 0, "adam_step_lr_dev: bad arguments");
    if (n == 0) return MPSR_OK;
    MPSR_REQUIRE(param && grad && m && v && lr_t_dev, "adam_step_lr_dev: null pointer");
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, mpsr::as_stream(stream), param, grad, m, v, n, 0.f,
                       beta1, beta2, eps, grad_scale, lr_t_dev);
    MPSR_CHECK_LAUNCH("adam_kernel");
    return MPSR_OK;
}

extern "C" void mpsr_debug_set_wgrad_target(int workgroups) { g_wgrad_target = workgroups; }
extern "C" void mpsr_debug_set_wgrad_grouped(int on) { g_wgrad_grouped = on; }
extern "C" void mpsr_debug_set_wgrad_min_steps(int steps) { g_wgrad_min_steps = steps; }
extern "C" void mpsr_debug_set_wgrad_direct(int on) { g_wgrad_direct = on; }
