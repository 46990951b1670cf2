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
//   * sweeps: one 1024-thread workgroup per cloud; the opposite cloud is staged through LDS as float4
//     (x, y, z, weight) tiles and read with wave-uniform ds_read_b128; each thread carries kPT own points;
//   * exp(x) is evaluated as exp2(x * log2 e) with log2 e folded into the level constant (v_exp_f32).
#include "common.h"

namespace {

constexpr int kLevels = 10;
constexpr int kSweepThreads = 1024;
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

__device__ __forceinline__ float sq3(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return dx * dx + dy * dy + dz * dz;
}

// Stage `cnt` points (xyz from `pts`, 4th lane from `wgt`) into the LDS tile.
__device__ __forceinline__ void stage_tile(float4 *tile, const float *__restrict__ pts, const float *__restrict__ wgt,
                                           int first, int cnt)
{
    for (int p = threadIdx.x; p < cnt; p += blockDim.x) {
        const float *s = pts + (size_t)(first + p) * 3;
        tile[p] = make_float4(s[0], s[1], s[2], wgt[first + p]);
    }
}

// State per cloud in `temp`: remL[n] remR[m] ratL[kLevels][n] ratR[kLevels][m].
__global__ __launch_bounds__(kSweepThreads) void approx_match_sweeps_kernel(int n, int m,
                                                                           const float *__restrict__ xyz1,
                                                                           const float *__restrict__ xyz2,
                                                                           float *__restrict__ temp)
{
    __shared__ float4 tile[kTile];
    const int cloud = blockIdx.x, tid = threadIdx.x;
    const float *p1 = xyz1 + (size_t)cloud * n * 3;
    const float *p2 = xyz2 + (size_t)cloud * m * 3;
    float *st = temp + (size_t)cloud * ((size_t)(n + m) * (1 + kLevels));
    float *remL = st, *remR = st + n;
    float *ratLall = st + n + m, *ratRall = ratLall + (size_t)kLevels * n;
    const float multiL = n >= m ? 1.f : (float)(m / n);
    const float multiR = n >= m ? (float)(n / m) : 1.f;
    for (int k = tid; k < n; k += kSweepThreads) remL[k] = multiL;
    for (int l = tid; l < m; l += kSweepThreads) remR[l] = multiR;
    __syncthreads();

    for (int lev = 0; lev < kLevels; ++lev) {
        const float lv = level_log2e(lev);
        float *ratL = ratLall + (size_t)lev * n;
        float *ratR = ratRall + (size_t)lev * m;

        // sweep 1: what each giver k could hand out, weighted by what receivers can still take
        for (int k0 = 0; k0 < n; k0 += kSweepThreads * kPT) {
            float x[kPT], y[kPT], z[kPT], s[kPT];
#pragma unroll
            for (int u = 0; u < kPT; ++u) {
                const int k = k0 + u * kSweepThreads + tid;
                const bool live = k < n;
                x[u] = live ? p1[3 * k] : 0.f;
                y[u] = live ? p1[3 * k + 1] : 0.f;
                z[u] = live ? p1[3 * k + 2] : 0.f;
                s[u] = 1e-9f;
            }
            for (int l0 = 0; l0 < m; l0 += kTile) {
                const int cnt = min(kTile, m - l0);
                __syncthreads();
                stage_tile(tile, p2, remR, l0, cnt);
                __syncthreads();
#pragma unroll 4
                for (int l = 0; l < cnt; ++l) {
                    const float4 t = tile[l];
#pragma unroll
                    for (int u = 0; u < kPT; ++u) s[u] += __builtin_amdgcn_exp2f(lv * sq3(t.x, t.y, t.z, x[u], y[u], z[u])) * t.w;
                }
            }
#pragma unroll
            for (int u = 0; u < kPT; ++u) {
                const int k = k0 + u * kSweepThreads + tid;
                if (k < n) ratL[k] = remL[k] / s[u];
            }
        }
        __syncthreads();

        // sweep 2: what each receiver l is offered; over-subscribed receivers scale down; receiver capacity update
        for (int l0 = 0; l0 < m; l0 += kSweepThreads * kPT) {
            float x[kPT], y[kPT], z[kPT], s[kPT];
#pragma unroll
            for (int u = 0; u < kPT; ++u) {
                const int l = l0 + u * kSweepThreads + tid;
                const bool live = l < m;
                x[u] = live ? p2[3 * l] : 0.f;
                y[u] = live ? p2[3 * l + 1] : 0.f;
                z[u] = live ? p2[3 * l + 2] : 0.f;
                s[u] = 0.f;
            }
            for (int k0 = 0; k0 < n; k0 += kTile) {
                const int cnt = min(kTile, n - k0);
                __syncthreads();
                stage_tile(tile, p1, ratL, k0, cnt);
                __syncthreads();
#pragma unroll 4
                for (int k = 0; k < cnt; ++k) {
                    const float4 t = tile[k];
#pragma unroll
                    for (int u = 0; u < kPT; ++u) s[u] += __builtin_amdgcn_exp2f(lv * sq3(x[u], y[u], z[u], t.x, t.y, t.z)) * t.w;
                }
            }
#pragma unroll
            for (int u = 0; u < kPT; ++u) {
                const int l = l0 + u * kSweepThreads + tid;
                if (l < m) {
                    const float rem = remR[l];
                    const float offered = s[u] * rem;
                    const float consumption = fminf(rem / (offered + 1e-9f), 1.0f);
                    ratR[l] = consumption * rem;
                    remR[l] = fmaxf(0.0f, rem - offered);
                }
            }
        }
        __syncthreads();

        // sweep 3: what each giver k actually hands out at this level; giver capacity update
        for (int k0 = 0; k0 < n; k0 += kSweepThreads * kPT) {
            float x[kPT], y[kPT], z[kPT], s[kPT], rl[kPT];
#pragma unroll
            for (int u = 0; u < kPT; ++u) {
                const int k = k0 + u * kSweepThreads + tid;
                const bool live = k < n;
                x[u] = live ? p1[3 * k] : 0.f;
                y[u] = live ? p1[3 * k + 1] : 0.f;
                z[u] = live ? p1[3 * k + 2] : 0.f;
                rl[u] = live ? ratL[k] : 0.f;
                s[u] = 0.f;
            }
            for (int l0 = 0; l0 < m; l0 += kTile) {
                const int cnt = min(kTile, m - l0);
                __syncthreads();
                stage_tile(tile, p2, ratR, l0, cnt);
                __syncthreads();
#pragma unroll 4
                for (int l = 0; l < cnt; ++l) {
                    const float4 t = tile[l];
#pragma unroll
                    for (int u = 0; u < kPT; ++u)
                        s[u] += __builtin_amdgcn_exp2f(lv * sq3(t.x, t.y, t.z, x[u], y[u], z[u])) * rl[u] * t.w;
                }
            }
#pragma unroll
            for (int u = 0; u < kPT; ++u) {
                const int k = k0 + u * kSweepThreads + tid;
                if (k < n) remL[k] = fmaxf(0.0f, remL[k] - s[u]);
            }
        }
        __syncthreads();
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

// cost[cloud] = sum_{l,k} |p2_l - p1_k| * match[l][k]; one workgroup per cloud streams its match slab once.
__global__ __launch_bounds__(1024) void match_cost_kernel(int n, int m, const float *__restrict__ xyz1,
                                                          const float *__restrict__ xyz2,
                                                          const float *__restrict__ match, float *__restrict__ out)
{
    __shared__ float4 tile[kTile];
    __shared__ float part[16];
    const int cloud = blockIdx.x, tid = threadIdx.x;
    const float *p1 = xyz1 + (size_t)cloud * n * 3;
    const float *p2 = xyz2 + (size_t)cloud * m * 3;
    const float *mt = match + (size_t)cloud * n * m;
    float acc = 0.f;
    for (int l0 = 0; l0 < m; l0 += kTile) {
        const int cnt = min(kTile, m - l0);
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
        if (tid == 0) out[cloud] = v;
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
    hipLaunchKernelGGL(approx_match_sweeps_kernel, dim3(b), dim3(kSweepThreads), 0, s, n, m, xyz1, xyz2, temp);
    MPSR_CHECK_LAUNCH("approx_match_sweeps_kernel");
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
    hipLaunchKernelGGL(match_cost_kernel, dim3(b), dim3(1024), 0, mpsr::as_stream(stream), n, m, xyz1, xyz2, match,
                       out);
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
