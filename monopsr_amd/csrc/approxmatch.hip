// Approximate Earth Mover's Distance (ops ApproxMatch / MatchCost / MatchCostGrad) for gfx950.
//
// Algorithm = the reference's device kernel (tf_ops/approxmatch/tf_approxmatch_g.cu:1-179): 10 annealing levels
// (level = -4^j, j = 7..-1, then 0), fp32 state, capacities multiL/multiR, three O(n*m) sweeps per level, and a
// match tensor laid out [m][n] per cloud as the (b, m, n) op output documents (tf_approxmatch.py:15-23).
//
// MI355X design.  The reference read-modify-writes the b*n*m match tensor once per level (10x); here the sweeps
// only update the O(n+m) state and record each level's giver/receiver ratios, and a second kernel emits
//     match[l][k] = sum_levels exp(level * |p2_l - p1_k|^2) * ratioL[level][k] * ratioR[level][l]
// (same terms, same summation order as the reference's accumulation) so match is written exactly once:
// b*n*m*4 bytes of HBM traffic instead of ~19x that, for 10 extra exponentials per pair.
//   * sweeps: one launch per sweep, 512 own points per 256-thread workgroup; the opposite cloud is staged through
//     LDS as float4 (x, y, z, weight) tiles and read with wave-uniform ds_read_b128; each thread carries kPT own
//     points;
//   * exp(x) is evaluated as exp2(x * log2 e) with log2 e folded into the level constant (v_exp_f32).
#include "common.h"

namespace {

constexpr int kLevels = 10;
constexpr int kSweepThreads = 256;
constexpr int kPT = 2;         // own points per thread in the sweeps
constexpr int kTile = 2048;    // opposite-cloud points per LDS tile (32 KiB)
constexpr float kLog2e = 1.4426950408889634f;

// level constant for annealing step lev (0..9), pre-multiplied by log2(e)
__device__ __forceinline__ float level_log2e(int lev)
{
    const int j = 7 - lev;  // 7..-2
    if (j == -2) return 0.f;
    return -exp2f(2.f * (float)j) * kLog2e;
}

using f32x2 = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ float sq3(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return dx * dx + dy * dy + dz * dz;
}

// Stage `cnt` points (xyz from `pts`, 4th lane from `wgt`, or the constant `wconst` when wgt == nullptr).
__device__ __forceinline__ void stage_tile(float4 *tile, const float *__restrict__ pts, const float *__restrict__ wgt,
                                           float wconst, int first, int cnt)
{
    for (int p = threadIdx.x; p < cnt; p += blockDim.x) {
        const float *s = pts + (size_t)(first + p) * 3;
        tile[p] = make_float4(s[0], s[1], s[2], wgt ? wgt[first + p] : wconst);
    }
}

// State per cloud in `temp`: remL[n] remR[m] ratL[kLevels][n] ratR[kLevels][m].
//
// One launch per sweep (3 per level, 30 in all): every sweep is independent per OWN point (sweeps 1 and 3 per giver
// k, sweep 2 per receiver l) and only needs the previous sweep finished, so the launch boundary is the only
// synchronisation and a cloud spreads over ceil(points / 512) workgroups instead of one -- the reference's
// 32-cloud evaluation batches fill the chip too.  grid = (ceil(own / (kSweepThreads * kPT)), b).
//   SWEEP 1: ratL[k] = remL[k] / (1e-9 + sum_l e(k,l) * remR[l])
//   SWEEP 2: s = remR[l] * sum_k e(k,l) * ratL[k];  ratR[l] = min(remR[l] / (s + 1e-9), 1) * remR[l];
//            remR[l] = max(0, remR[l] - s)
//   SWEEP 3: remL[k] = max(0, remL[k] - ratL[k] * sum_l e(k,l) * ratR[l])
// At level 0 the capacities are the constants multiL / multiR (no initialisation pass).
template <int SWEEP>
__global__ __launch_bounds__(kSweepThreads) void approx_match_sweep_kernel(int n, int m,
                                                                          const float *__restrict__ xyz1,
                                                                          const float *__restrict__ xyz2,
                                                                          float *__restrict__ temp, int lev)
{
    __shared__ float4 tile[kTile];
    const int cloud = blockIdx.y, tid = threadIdx.x;
    const float *p1 = xyz1 + (size_t)cloud * n * 3;
    const float *p2 = xyz2 + (size_t)cloud * m * 3;
    float *st = temp + (size_t)cloud * ((size_t)(n + m) * (1 + kLevels));
    float *remL = st, *remR = st + n;
    float *ratL = st + n + m + (size_t)lev * n;
    float *ratR = st + n + m + (size_t)kLevels * n + (size_t)lev * m;
    const float multiL = n >= m ? 1.f : (float)(m / n);
    const float multiR = n >= m ? (float)(n / m) : 1.f;
    const bool first = lev == 0;
    const float lv = level_log2e(lev);
    constexpr bool kOwnIsL = SWEEP != 2;           // own points come from cloud 1 (givers) in sweeps 1 and 3
    const int nown = kOwnIsL ? n : m, nother = kOwnIsL ? m : n;
    const float *own = kOwnIsL ? p1 : p2, *other = kOwnIsL ? p2 : p1;
    const float *wgt = SWEEP == 1 ? (first ? nullptr : remR) : (SWEEP == 2 ? ratL : ratR);

    const int base = blockIdx.x * (kSweepThreads * kPT);
    // the two own points of a thread live in the halves of 2-wide vectors: the distance / weighting arithmetic then
    // compiles to packed fp32 instructions (v_pk_add/mul/fma_f32: two points per VALU slot); only the two exp2 stay
    // scalar (quarter-rate transcendental unit)
    static_assert(kPT == 2, "packed sweep arithmetic pairs two own points per thread");
    f32x2 x, y, z, sv;
#pragma unroll
    for (int u = 0; u < kPT; ++u) {
        const int i = base + u * kSweepThreads + tid;
        const bool live = i < nown;
        x[u] = live ? own[3 * i] : 0.f;
        y[u] = live ? own[3 * i + 1] : 0.f;
        z[u] = live ? own[3 * i + 2] : 0.f;
        sv[u] = SWEEP == 1 ? 1e-9f : 0.f;
    }
    for (int o0 = 0; o0 < nother; o0 += kTile) {
        const int cnt = min(kTile, nother - o0);
        __syncthreads();
        stage_tile(tile, other, wgt, multiR, o0, cnt);
        __syncthreads();
#pragma unroll 4
        for (int o = 0; o < cnt; ++o) {
            const float4 t = tile[o];
            // distance evaluated as (other - own) for givers, (own - other) for receivers: the reference's operand
            // order (x2 - x1); squares are identical either way
            const f32x2 dx = kOwnIsL ? t.x - x : x - t.x, dy = kOwnIsL ? t.y - y : y - t.y,
                        dz = kOwnIsL ? t.z - z : z - t.z;
            const f32x2 d2 = dx * dx + dy * dy + dz * dz;
            const f32x2 a = lv * d2;
            f32x2 e;
            e[0] = __builtin_amdgcn_exp2f(a[0]);
            e[1] = __builtin_amdgcn_exp2f(a[1]);
            sv += e * t.w;
        }
    }
    float s[kPT] = {sv[0], sv[1]};
#pragma unroll
    for (int u = 0; u < kPT; ++u) {
        const int i = base + u * kSweepThreads + tid;
        if (i >= nown) continue;
        if (SWEEP == 1) {
            ratL[i] = (first ? multiL : remL[i]) / s[u];
        } else if (SWEEP == 2) {
            const float rem = first ? multiR : remR[i];
            const float offered = s[u] * rem;
            const float consumption = fminf(rem / (offered + 1e-9f), 1.0f);
            ratR[i] = consumption * rem;
            remR[i] = fmaxf(0.0f, rem - offered);
        } else {
            // the reference multiplies every term by ratL[k] inside the sum; factoring it out changes the rounding
            // of the sum by < 1 ulp per term and stays far inside the 1e-3 bar
            const float given = s[u] * ratL[i];
            remL[i] = fmaxf(0.0f, (first ? multiL : remL[i]) - given);
        }
    }
}

constexpr int kEmitRows = 64;  // receiver rows (l) per workgroup in the emit kernel

// match[l][k] for one cloud; grid (ceil(n/256), ceil(m/kEmitRows), b).
__global__ __launch_bounds__(256) void approx_match_emit_kernel(int n, int m, const float *__restrict__ xyz1,
                                                                const float *__restrict__ xyz2,
                                                                const float *__restrict__ temp,
                                                                float *__restrict__ match)
{
    __shared__ float rowp[kEmitRows][4];            // x, y, z of receiver l
    __shared__ float rowr[kEmitRows][kLevels + 2];  // its ratio per level (padded)
    const int cloud = blockIdx.z, tid = threadIdx.x;
    const int k = blockIdx.x * 256 + tid;
    const int l0 = blockIdx.y * kEmitRows;
    const int rows = min(kEmitRows, m - l0);
    const float *p1 = xyz1 + (size_t)cloud * n * 3;
    const float *p2 = xyz2 + (size_t)cloud * m * 3;
    const float *st = temp + (size_t)cloud * ((size_t)(n + m) * (1 + kLevels));
    const float *ratLall = st + n + m, *ratRall = ratLall + (size_t)kLevels * n;
    for (int i = tid; i < rows * 3; i += 256) rowp[i / 3][i % 3] = p2[(size_t)l0 * 3 + i];
    for (int i = tid; i < rows * kLevels; i += 256) {
        const int r = i / kLevels, lev = i % kLevels;
        rowr[r][lev] = ratRall[(size_t)lev * m + l0 + r];
    }
    float lvl[kLevels], rl[kLevels];
    const bool live = k < n;
    const float x = live ? p1[3 * k] : 0.f, y = live ? p1[3 * k + 1] : 0.f, z = live ? p1[3 * k + 2] : 0.f;
#pragma unroll
    for (int lev = 0; lev < kLevels; ++lev) {
        lvl[lev] = level_log2e(lev);
        rl[lev] = live ? ratLall[(size_t)lev * n + k] : 0.f;
    }
    __syncthreads();
    if (!live) return;
    float *out = match + ((size_t)cloud * m + l0) * n + k;
    for (int r = 0; r < rows; ++r) {
        const float d2 = sq3(rowp[r][0], rowp[r][1], rowp[r][2], x, y, z);
        float acc = 0.f;
#pragma unroll
        for (int lev = 0; lev < kLevels; ++lev) acc += __builtin_amdgcn_exp2f(lvl[lev] * d2) * rl[lev] * rowr[r][lev];
        out[(size_t)r * n] = acc;
    }
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// cost[cloud] = sum_{l,k} |p2_l - p1_k| * match[l][k]; the match slab is streamed once.  grid (b, slices): a
// workgroup takes `rows` receiver rows of one cloud; with more than one slice per cloud the partial sums meet in an
// fp32 atomic on the pre-zeroed output.
__global__ __launch_bounds__(1024) void match_cost_kernel(int n, int m, int rows, const float *__restrict__ xyz1,
                                                          const float *__restrict__ xyz2,
                                                          const float *__restrict__ match, float *__restrict__ out,
                                                          int atomic)
{
    __shared__ float4 tile[kTile];
    __shared__ float part[16];
    const int cloud = blockIdx.x, tid = threadIdx.x;
    const float *p1 = xyz1 + (size_t)cloud * n * 3;
    const float *p2 = xyz2 + (size_t)cloud * m * 3;
    const float *mt = match + (size_t)cloud * n * m;
    const int lbeg = blockIdx.y * rows, lend = min(m, lbeg + rows);
    float acc = 0.f;
    for (int l0 = lbeg; l0 < lend; l0 += kTile) {
        const int cnt = min(kTile, lend - l0);
        __syncthreads();
        for (int p = tid; p < cnt; p += 1024) {
            const float *s = p2 + (size_t)(l0 + p) * 3;
            tile[p] = make_float4(s[0], s[1], s[2], 0.f);
        }
        __syncthreads();
        for (int k = tid; k < n; k += 1024) {
            const float x = p1[3 * k], y = p1[3 * k + 1], z = p1[3 * k + 2];
            const float *col = mt + (size_t)l0 * n + k;
#pragma unroll 8
            for (int l = 0; l < cnt; ++l) {
                const float4 t = tile[l];
                acc += sqrtf(sq3(t.x, t.y, t.z, x, y, z)) * col[(size_t)l * n];
            }
        }
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) part[tid >> 6] = acc;
    __syncthreads();
    if (tid < 64) {
        float v = tid < 16 ? part[tid] : 0.f;
        v = wave_sum(v);
        if (tid == 0) {
            if (atomic) atomicAdd(&out[cloud], v);
            else out[cloud] = v;
        }
    }
}

// grad1[k] = sum_l match[l][k] * (p1_k - p2_l) / max(|.|, 1e-10); thread per k, grid (ceil(n/256), b).
__global__ __launch_bounds__(256) void match_cost_grad1_kernel(int n, int m, const float *__restrict__ xyz1,
                                                               const float *__restrict__ xyz2,
                                                               const float *__restrict__ match,
                                                               float *__restrict__ grad1)
{
    __shared__ float4 tile[kTile];
    const int cloud = blockIdx.y, tid = threadIdx.x;
    const int k = blockIdx.x * 256 + tid;
    const bool live = k < n;
    const float *p1 = xyz1 + (size_t)cloud * n * 3;
    const float *p2 = xyz2 + (size_t)cloud * m * 3;
    const float *mt = match + (size_t)cloud * n * m;
    const float x = live ? p1[3 * k] : 0.f, y = live ? p1[3 * k + 1] : 0.f, z = live ? p1[3 * k + 2] : 0.f;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int l0 = 0; l0 < m; l0 += kTile) {
        const int cnt = min(kTile, m - l0);
        __syncthreads();
        for (int p = tid; p < cnt; p += 256) {
            const float *s = p2 + (size_t)(l0 + p) * 3;
            tile[p] = make_float4(s[0], s[1], s[2], 0.f);
        }
        __syncthreads();
        if (live) {
            const float *col = mt + (size_t)l0 * n + k;
#pragma unroll 8
            for (int l = 0; l < cnt; ++l) {
                const float4 t = tile[l];
                const float dx = x - t.x, dy = y - t.y, dz = z - t.z;
                const float s = col[(size_t)l * n] * rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-20f));
                gx += dx * s;
                gy += dy * s;
                gz += dz * s;
            }
        }
    }
    if (live) {
        float *g = grad1 + ((size_t)cloud * n + k) * 3;
        g[0] = gx;
        g[1] = gy;
        g[2] = gz;
    }
}

// grad2[l] = sum_k match[l][k] * (p2_l - p1_k) / max(|.|, 1e-10); one wave per row l (coalesced row reads),
// 4 rows per workgroup, grid (ceil(m/4), b).
__global__ __launch_bounds__(256) void match_cost_grad2_kernel(int n, int m, const float *__restrict__ xyz1,
                                                               const float *__restrict__ xyz2,
                                                               const float *__restrict__ match,
                                                               float *__restrict__ grad2)
{
    const int cloud = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= m) return;
    const float *p1 = xyz1 + (size_t)cloud * n * 3;
    const float *p2 = xyz2 + ((size_t)cloud * m + l) * 3;
    const float *row = match + ((size_t)cloud * m + l) * n;
    const float x = p2[0], y = p2[1], z = p2[2];
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int k = lane; k < n; k += 64) {
        const float dx = x - p1[3 * k], dy = y - p1[3 * k + 1], dz = z - p1[3 * k + 2];
        const float s = row[k] * rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-20f));
        gx += dx * s;
        gy += dy * s;
        gz += dz * s;
    }
    gx = wave_sum(gx);
    gy = wave_sum(gy);
    gz = wave_sum(gz);
    if (lane == 0) {
        float *g = grad2 + ((size_t)cloud * m + l) * 3;
        g[0] = gx;
        g[1] = gy;
        g[2] = gz;
    }
}

int check_emd_args(const char *op, int b, int n, int m)
{
    MPSR_REQUIRE(b >= 0 && n >= 0 && m >= 0, "%s: negative size (b=%d n=%d m=%d)", op, b, n, m);
    if (b == 0) return MPSR_OK;
    MPSR_REQUIRE(n > 0 && m > 0, "%s: both clouds need at least one point (n=%d m=%d)", op, n, m);
    MPSR_REQUIRE(b <= 65535, "%s: batch %d exceeds 65535", op, b);
    return MPSR_OK;
}

}  // namespace

extern "C" size_t mpsr_approx_match_temp_floats(int b, int n, int m)
{
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    return (size_t)b * ((size_t)(n + m) * (1 + kLevels));
}

extern "C" int mpsr_approx_match(int b, int n, int m, const float *xyz1, const float *xyz2, float *match,
                                 float *temp, mpsr_stream_t stream)
{
    if (int rc = check_emd_args("approx_match", b, n, m)) return rc;
    if (b == 0) return MPSR_OK;
    MPSR_REQUIRE(xyz1 && xyz2 && match && temp, "approx_match: null pointer");
    hipStream_t s = mpsr::as_stream(stream);
    const dim3 gl(mpsr::ceil_div(n, kSweepThreads * kPT), b), gr(mpsr::ceil_div(m, kSweepThreads * kPT), b);
    for (int lev = 0; lev < kLevels; ++lev) {
        hipLaunchKernelGGL(approx_match_sweep_kernel<1>, gl, dim3(kSweepThreads), 0, s, n, m, xyz1, xyz2, temp, lev);
        hipLaunchKernelGGL(approx_match_sweep_kernel<2>, gr, dim3(kSweepThreads), 0, s, n, m, xyz1, xyz2, temp, lev);
        hipLaunchKernelGGL(approx_match_sweep_kernel<3>, gl, dim3(kSweepThreads), 0, s, n, m, xyz1, xyz2, temp, lev);
    }
    MPSR_CHECK_LAUNCH("approx_match_sweep_kernel");
    dim3 grid(mpsr::ceil_div(n, 256), mpsr::ceil_div(m, kEmitRows), b);
    MPSR_REQUIRE(grid.y <= 65535, "approx_match: m=%d too large", m);
    hipLaunchKernelGGL(approx_match_emit_kernel, grid, dim3(256), 0, s, n, m, xyz1, xyz2, temp, match);
    MPSR_CHECK_LAUNCH("approx_match_emit_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_match_cost(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                               float *out, mpsr_stream_t stream)
{
    if (int rc = check_emd_args("match_cost", b, n, m)) return rc;
    if (b == 0) return MPSR_OK;
    MPSR_REQUIRE(xyz1 && xyz2 && match && out, "match_cost: null pointer");
    hipStream_t s = mpsr::as_stream(stream);
    int slices = 1024 / b;  // aim at ~4 workgroups per CU
    if (slices < 1) slices = 1;
    if (slices > 32) slices = 32;
    if (slices > m) slices = m;
    const int rows = mpsr::ceil_div(m, slices);
    slices = mpsr::ceil_div(m, rows);
    if (slices > 1) MPSR_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float) * b, s));
    hipLaunchKernelGGL(match_cost_kernel, dim3(b, slices), dim3(1024), 0, s, n, m, rows, xyz1, xyz2, match, out,
                       slices > 1 ? 1 : 0);
    MPSR_CHECK_LAUNCH("match_cost_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_match_cost_grad(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                                    float *grad1, float *grad2, mpsr_stream_t stream)
{
    if (int rc = check_emd_args("match_cost_grad", b, n, m)) return rc;
    if (b == 0) return MPSR_OK;
    MPSR_REQUIRE(xyz1 && xyz2 && match && grad1 && grad2, "match_cost_grad: null pointer");
    hipStream_t s = mpsr::as_stream(stream);
    hipLaunchKernelGGL(match_cost_grad1_kernel, dim3(mpsr::ceil_div(n, 256), b), dim3(256), 0, s, n, m, xyz1, xyz2,
                       match, grad1);
    MPSR_CHECK_LAUNCH("match_cost_grad1_kernel");
    hipLaunchKernelGGL(match_cost_grad2_kernel, dim3(mpsr::ceil_div(m, 4), b), dim3(256), 0, s, n, m, xyz1, xyz2,
                       match, grad2);
    MPSR_CHECK_LAUNCH("match_cost_grad2_kernel");
    return MPSR_OK;
}
