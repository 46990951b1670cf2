// Chamfer nearest-neighbour search (op NnDistance) and its gradient (op NnDistanceGrad) for gfx950.
//
// Semantics follow the reference's CPU kernel bit for bit (tf_ops/nn_distance/tf_nndistance.cpp:21-43): float
// differences, float products, float sums associated (dx*dx + dy*dy) + dz*dz with NO fused multiply-add, first
// candidate seeds, strict '<' replaces (lowest index wins ties, NaN candidates never win, a NaN first candidate
// sticks).  The file is compiled with -ffp-contract=off and repeats that as a pragma.
//
// Forward design (VALU-bound: 9 lane-ops per point pair):
//   * one launch covers both directions (blockIdx.z) and all clouds (blockIdx.y);
//   * the searched cloud is staged through LDS as float4 tiles and read back with wave-uniform (broadcast)
//     ds_read_b128; each thread carries kQPT query points so one LDS read feeds kQPT distance evaluations;
//   * the inner loop tracks only the running minimum per 8-point chunk (v_min, no index bookkeeping); the chunk
//     that first lowered the minimum is remembered and re-scanned once at the end to recover the exact lowest
//     index.  Re-evaluating a distance reproduces the same bits, so value and index equal the sequential scan.
// Backward design: one workgroup per cloud accumulates both gradients in LDS (direct stores for a point's own
// term, ds_add_f32 for the scattered term) and writes each output once; clouds too large for LDS take a
// global-atomic path.
#include <mutex>

#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int kThreads = 256;
constexpr int kQPT = 4;      // query points per thread (one LDS broadcast read feeds kQPT distance evaluations)
constexpr int kTile = 1024;  // searched points per LDS tile (16 KiB as float4)
constexpr int kChunk = 8;    // points per min-tracking chunk

__device__ __forceinline__ float sqdist(float rx, float ry, float rz, float qx, float qy, float qz)
{
    const float dx = rx - qx;
    const float dy = ry - qy;
    const float dz = rz - qz;
    return (dx * dx + dy * dy) + dz * dz;
}

using f32x2 = __attribute__((ext_vector_type(2))) float;

__global__ __launch_bounds__(kThreads) void nn_search_kernel(int n, const float *__restrict__ xyz1, int m,
                                                             const float *__restrict__ xyz2,
                                                             float *__restrict__ dist1, int *__restrict__ idx1,
                                                             float *__restrict__ dist2, int *__restrict__ idx2)
{
    __shared__ float4 tile[kTile];
    const int dir = blockIdx.z;
    const int cloud = blockIdx.y;
    const int nq = dir ? m : n;
    const int nr = dir ? n : m;
    const int qbase = blockIdx.x * (kThreads * kQPT);
    if (qbase >= nq) return;  // block-uniform: the grid is sized for max(n, m)
    const float *q = (dir ? xyz2 : xyz1) + (size_t)cloud * nq * 3;
    const float *r = (dir ? xyz1 : xyz2) + (size_t)cloud * nr * 3;
    float *dist = (dir ? dist2 : dist1) + (size_t)cloud * nq;
    int *idx = (dir ? idx2 : idx1) + (size_t)cloud * nq;
    const int tid = threadIdx.x;

    float qx[kQPT], qy[kQPT], qz[kQPT], best[kQPT];
    int bchunk[kQPT];
#pragma unroll
    for (int u = 0; u < kQPT; ++u) {
        const int j = qbase + u * kThreads + tid;
        const bool live = j < nq;
        qx[u] = live ? q[3 * j] : 0.f;
        qy[u] = live ? q[3 * j + 1] : 0.f;
        qz[u] = live ? q[3 * j + 2] : 0.f;
        best[u] = __builtin_inff();
        bchunk[u] = 0;
    }

    for (int t0 = 0; t0 < nr; t0 += kTile) {
        const int cnt = min(kTile, nr - t0);
        const int cntp = (cnt + kChunk - 1) / kChunk * kChunk;  // pad to whole chunks with +inf points
        __syncthreads();
        for (int p = tid; p < cntp; p += kThreads) {
            float4 v;
            if (p < cnt) {
                const float *s = r + (size_t)(t0 + p) * 3;
                v = make_float4(s[0], s[1], s[2], 0.f);
            } else {
                v = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.f);
            }
            tile[p] = v;
        }
        __syncthreads();
        // the thread's two queries sit in the halves of 2-wide vectors: subtract / square / add issue as packed fp32
        // (v_pk_add_f32 / v_pk_mul_f32 are IEEE-exact and nothing is contracted, so every distance keeps the bits of
        // the scalar expression (dx*dx + dy*dy) + dz*dz); only the running minimum stays per component
        static_assert(kQPT % 2 == 0, "packed search arithmetic pairs the queries of a thread");
        f32x2 vqx[kQPT / 2], vqy[kQPT / 2], vqz[kQPT / 2];
#pragma unroll
        for (int v = 0; v < kQPT / 2; ++v) {
            vqx[v] = f32x2{qx[2 * v], qx[2 * v + 1]};
            vqy[v] = f32x2{qy[2 * v], qy[2 * v + 1]};
            vqz[v] = f32x2{qz[2 * v], qz[2 * v + 1]};
        }
        for (int k0 = 0; k0 < cntp; k0 += kChunk) {
            f32x2 cmv[kQPT / 2];
#pragma unroll
            for (int v = 0; v < kQPT / 2; ++v) cmv[v] = f32x2{__builtin_inff(), __builtin_inff()};
#pragma unroll
            for (int s = 0; s < kChunk; ++s) {
                const float4 rp = tile[k0 + s];
#pragma unroll
                for (int v = 0; v < kQPT / 2; ++v) {
                    const f32x2 dx = rp.x - vqx[v], dy = rp.y - vqy[v], dz = rp.z - vqz[v];
                    cmv[v] = __builtin_elementwise_min(cmv[v], (dx * dx + dy * dy) + dz * dz);
                }
            }
            float cm[kQPT];
#pragma unroll
            for (int v = 0; v < kQPT / 2; ++v) {
                cm[2 * v] = cmv[v][0];
                cm[2 * v + 1] = cmv[v][1];
            }
#pragma unroll
            for (int u = 0; u < kQPT; ++u) {
                if (cm[u] < best[u]) {
                    best[u] = cm[u];
                    bchunk[u] = t0 + k0;
                }
            }
        }
    }

#pragma unroll
    for (int u = 0; u < kQPT; ++u) {
        const int j = qbase + u * kThreads + tid;
        if (j >= nq) continue;
        const float d0 = sqdist(r[0], r[1], r[2], qx[u], qy[u], qz[u]);
        float bd = best[u];
        int bi = bchunk[u];
        if (d0 != d0) {  // NaN seed: nothing compares below it
            bd = d0;
            bi = 0;
        } else {
            for (int s = 0; s < kChunk; ++s) {
                const int k = bchunk[u] + s;
                if (k >= nr) break;
                const float *rp = r + (size_t)k * 3;
                if (sqdist(rp[0], rp[1], rp[2], qx[u], qy[u], qz[u]) == bd) {
                    bi = k;
                    break;
                }
            }
        }
        dist[j] = bd;
        idx[j] = bi;
    }
}

// One workgroup per cloud; acc = [3n floats for cloud 1 | 3m floats for cloud 2] in LDS.
__global__ __launch_bounds__(1024) void nn_grad_lds_kernel(int n, const float *__restrict__ xyz1, int m,
                                                           const float *__restrict__ xyz2,
                                                           const float *__restrict__ gd1, const int *__restrict__ idx1,
                                                           const float *__restrict__ gd2, const int *__restrict__ idx2,
                                                           float *__restrict__ gout1, float *__restrict__ gout2)
{
    extern __shared__ __attribute__((aligned(16))) float acc[];
    float *a1 = acc, *a2 = acc + 3 * n;
    const int cloud = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const float *p1 = xyz1 + (size_t)cloud * n * 3, *p2 = xyz2 + (size_t)cloud * m * 3;
    gd1 += (size_t)cloud * n;
    idx1 += (size_t)cloud * n;
    gd2 += (size_t)cloud * m;
    idx2 += (size_t)cloud * m;
    for (int i = tid; i < 3 * m; i += nt) a2[i] = 0.f;
    __syncthreads();
    // pass A: cloud-1 points; own term is the first contribution to a1[j] (reference order), scatter into a2
    for (int j = tid; j < n; j += nt) {
        const int t = idx1[j];
        const float g = gd1[j] * 2;
        float ex = 0.f, ey = 0.f, ez = 0.f;
        if ((unsigned)t < (unsigned)m) {
            ex = g * (p1[3 * j] - p2[3 * t]);
            ey = g * (p1[3 * j + 1] - p2[3 * t + 1]);
            ez = g * (p1[3 * j + 2] - p2[3 * t + 2]);
            atomicAdd(&a2[3 * t], -ex);
            atomicAdd(&a2[3 * t + 1], -ey);
            atomicAdd(&a2[3 * t + 2], -ez);
        }
        a1[3 * j] = ex;
        a1[3 * j + 1] = ey;
        a1[3 * j + 2] = ez;
    }
    __syncthreads();
    // pass B: cloud-2 points; own term lands after the scattered ones (reference order), scatter into a1
    for (int j = tid; j < m; j += nt) {
        const int t = idx2[j];
        const float g = gd2[j] * 2;
        if ((unsigned)t < (unsigned)n) {
            const float ex = g * (p2[3 * j] - p1[3 * t]);
            const float ey = g * (p2[3 * j + 1] - p1[3 * t + 1]);
            const float ez = g * (p2[3 * j + 2] - p1[3 * t + 2]);
            a2[3 * j] += ex;
            a2[3 * j + 1] += ey;
            a2[3 * j + 2] += ez;
            atomicAdd(&a1[3 * t], -ex);
            atomicAdd(&a1[3 * t + 1], -ey);
            atomicAdd(&a1[3 * t + 2], -ez);
        }
    }
    __syncthreads();
    float *o1 = gout1 + (size_t)cloud * n * 3, *o2 = gout2 + (size_t)cloud * m * 3;
    for (int i = tid; i < 3 * n; i += nt) o1[i] = a1[i];
    for (int i = tid; i < 3 * m; i += nt) o2[i] = a2[i];
}

// Large-cloud path, one pass per launch.  first_pass: own term is a plain store (sole writer), scatter target
// was zeroed by the caller.  second pass: own term is a plain read-modify-write (scatter into it finished with
// the previous launch), scatter target receives atomics.
__global__ __launch_bounds__(256) void nn_grad_global_kernel(int np, const float *__restrict__ p, int no,
                                                             const float *__restrict__ o,
                                                             const float *__restrict__ gd, const int *__restrict__ nn,
                                                             float *__restrict__ gp, float *__restrict__ go,
                                                             int first_pass)
{
    const int cloud = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= np) return;
    p += (size_t)cloud * np * 3;
    o += (size_t)cloud * no * 3;
    gp += (size_t)cloud * np * 3;
    go += (size_t)cloud * no * 3;
    const int t = nn[(size_t)cloud * np + j];
    const float g = gd[(size_t)cloud * np + j] * 2;
    float ex = 0.f, ey = 0.f, ez = 0.f;
    if ((unsigned)t < (unsigned)no) {
        ex = g * (p[3 * j] - o[3 * t]);
        ey = g * (p[3 * j + 1] - o[3 * t + 1]);
        ez = g * (p[3 * j + 2] - o[3 * t + 2]);
        atomicAdd(&go[3 * t], -ex);
        atomicAdd(&go[3 * t + 1], -ey);
        atomicAdd(&go[3 * t + 2], -ez);
    }
    if (first_pass) {
        gp[3 * j] = ex;
        gp[3 * j + 1] = ey;
        gp[3 * j + 2] = ez;
    } else {
        gp[3 * j] += ex;
        gp[3 * j + 1] += ey;
        gp[3 * j + 2] += ez;
    }
}

constexpr size_t kGradLdsLimit = 144 * 1024;  // leave headroom below the 160 KiB/CU LDS

}  // namespace

extern "C" int mpsr_nn_distance_fwd(int b, int n, const float *xyz1, int m, const float *xyz2, float *dist1,
                                    int *idx1, float *dist2, int *idx2, mpsr_stream_t stream)
{
    MPSR_REQUIRE(b >= 0 && n >= 0 && m >= 0, "nn_distance: negative size (b=%d n=%d m=%d)", b, n, m);
    if (b == 0 || (n == 0 && m == 0)) return MPSR_OK;
    MPSR_REQUIRE(n > 0 && m > 0, "nn_distance: both clouds need at least one point (n=%d m=%d)", n, m);
    MPSR_REQUIRE(xyz1 && xyz2 && dist1 && idx1 && dist2 && idx2, "nn_distance: null pointer");
    MPSR_REQUIRE(b <= 65535, "nn_distance: batch %d exceeds 65535", b);
    const int big = n > m ? n : m;
    dim3 grid(mpsr::ceil_div(big, kThreads * kQPT), b, 2);
    hipLaunchKernelGGL(nn_search_kernel, grid, dim3(kThreads), 0, mpsr::as_stream(stream), n, xyz1, m, xyz2, dist1,
                       idx1, dist2, idx2);
    MPSR_CHECK_LAUNCH("nn_search_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_nn_distance_bwd(int b, int n, const float *xyz1, int m, const float *xyz2,
                                    const float *grad_dist1, const int *idx1, const float *grad_dist2,
                                    const int *idx2, float *grad_xyz1, float *grad_xyz2, mpsr_stream_t stream)
{
    MPSR_REQUIRE(b >= 0 && n >= 0 && m >= 0, "nn_distance_grad: negative size (b=%d n=%d m=%d)", b, n, m);
    if (b == 0 || (n == 0 && m == 0)) return MPSR_OK;
    MPSR_REQUIRE(n > 0 && m > 0, "nn_distance_grad: both clouds need at least one point (n=%d m=%d)", n, m);
    MPSR_REQUIRE(xyz1 && xyz2 && grad_dist1 && idx1 && grad_dist2 && idx2 && grad_xyz1 && grad_xyz2,
                 "nn_distance_grad: null pointer");
    hipStream_t s = mpsr::as_stream(stream);
    const size_t lds = sizeof(float) * 3 * ((size_t)n + m);
    if (lds <= kGradLdsLimit) {
        // raise the kernel's dynamic-LDS cap (per call: the attribute belongs to the current device, the call is
        // cheap and idempotent, and a process may use several GPUs)
        MPSR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(nn_grad_lds_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGradLdsLimit));
        const int big = n > m ? n : m;
        const int threads = big >= 1024 ? 1024 : (big > 256 ? 512 : 256);
        hipLaunchKernelGGL(nn_grad_lds_kernel, dim3(b), dim3(threads), lds, s, n, xyz1, m, xyz2, grad_dist1, idx1,
                           grad_dist2, idx2, grad_xyz1, grad_xyz2);
        MPSR_CHECK_LAUNCH("nn_grad_lds_kernel");
        return MPSR_OK;
    }
    MPSR_REQUIRE(b <= 65535, "nn_distance_grad: batch %d exceeds 65535", b);
    MPSR_CHECK_HIP(hipMemsetAsync(grad_xyz2, 0, sizeof(float) * 3 * (size_t)b * m, s));
    hipLaunchKernelGGL(nn_grad_global_kernel, dim3(mpsr::ceil_div(n, 256), b), dim3(256), 0, s, n, xyz1, m, xyz2,
                       grad_dist1, idx1, grad_xyz1, grad_xyz2, 1);
    MPSR_CHECK_LAUNCH("nn_grad_global_kernel(A)");
    hipLaunchKernelGGL(nn_grad_global_kernel, dim3(mpsr::ceil_div(m, 256), b), dim3(256), 0, s, m, xyz2, n, xyz1,
                       grad_dist2, idx2, grad_xyz2, grad_xyz1, 0);
    MPSR_CHECK_LAUNCH("nn_grad_global_kernel(B)");
    return MPSR_OK;
}
