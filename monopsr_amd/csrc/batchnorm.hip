// Training-mode batch normalisation of the map decoder (net_builder.py:76-87: slim.conv2d(..., normalizer_fn=
// slim.batch_norm, normalizer_params={'is_training': is_training}) -- batch statistics over (N, H, W), no scale,
// epsilon 1e-3, followed by ReLU).  Inference folds the moving statistics into the convolution (core/weights.py);
// these kernels are what a training step needs instead.  All of them stream an (M, C) activation once, float4 per
// thread; per-channel reductions are per-thread float partials -> workgroup sums in double -> one fp64 atomic per
// channel per workgroup.  The reducing kernels run as ONE 1024-thread workgroup per CU with FOUR rows per thread in
// flight: with one 16-byte load outstanding per thread and 512 workgroups of 256 they streamed at 2.3-3.6 TB/s where
// the elementwise passes next to them reach 6+ (r05: 1.13 of the training step's 51 ms on the four decoder layers), and
// more workgroups cost more in same-address fp64 atomics than they gain (2048 x 256: 97 -> 197 us on bn_stats).
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRThreads = 1024;  // the reducing kernels
constexpr int kMaxBlocks = 256;

struct RowMap {  // thread -> (float4 column group, first row, row stride) for an (M, C) matrix, C % 4 == 0, C <= 1024
    int groups, rstride, cg, r0;
    __device__ RowMap(int C) : groups(C >> 2), rstride(kRThreads / (C >> 2)), cg(threadIdx.x % (C >> 2)),
                               r0(threadIdx.x / (C >> 2)) {}
};

// column sums of two per-thread float4 partials -> fp64 atomics into a[], b[]
__device__ __forceinline__ void reduce_columns(const RowMap &m, float4 sa, float4 sb, double *a, double *b)
{
    __shared__ float4 pa[kRThreads], pb[kRThreads];
    pa[threadIdx.x] = sa;
    pb[threadIdx.x] = sb;
    __syncthreads();
    if ((int)threadIdx.x < m.groups) {
        double ax = 0, ay = 0, az = 0, aw = 0, bx = 0, by = 0, bz = 0, bw = 0;
        for (int r = 0; r < m.rstride; ++r) {
            const float4 u = pa[r * m.groups + threadIdx.x], v = pb[r * m.groups + threadIdx.x];
            ax += u.x; ay += u.y; az += u.z; aw += u.w;
            bx += v.x; by += v.y; bz += v.z; bw += v.w;
        }
        const int c = 4 * threadIdx.x;
        unsafeAtomicAdd(&a[c], ax); unsafeAtomicAdd(&a[c + 1], ay); unsafeAtomicAdd(&a[c + 2], az);
        unsafeAtomicAdd(&a[c + 3], aw);
        unsafeAtomicAdd(&b[c], bx); unsafeAtomicAdd(&b[c + 1], by); unsafeAtomicAdd(&b[c + 2], bz);
        unsafeAtomicAdd(&b[c + 3], bw);
    }
}

// sum[c] += sum_m z, sumsq[c] += sum_m (z - shift[c])^2 ... shift = z of row 0 keeps the variance well conditioned
__global__ __launch_bounds__(kRThreads) void bn_stats_kernel(const float *__restrict__ z, long long M, int C,
                                                            long long rows_per_block, double *__restrict__ sum,
                                                            double *__restrict__ sumsq)
{
    const RowMap m(C);
    const long long m0 = (long long)blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    if (m.r0 < m.rstride) {
        const float4 sh = reinterpret_cast<const float4 *>(z)[m.cg];  // row 0 of this column group
        const float4 *zp = reinterpret_cast<const float4 *>(z);
        auto acc = [&](float4 v) __attribute__((always_inline)) {
            v.x -= sh.x; v.y -= sh.y; v.z -= sh.z; v.w -= sh.w;
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
        };
        long long r = m0 + m.r0;
        const long long st = m.rstride;
        for (; r + 3 * st < m1; r += 4 * st) {
            const float4 v0 = zp[(size_t)r * m.groups + m.cg], v1 = zp[(size_t)(r + st) * m.groups + m.cg],
                         v2 = zp[(size_t)(r + 2 * st) * m.groups + m.cg], v3 = zp[(size_t)(r + 3 * st) * m.groups + m.cg];
            acc(v0); acc(v1); acc(v2); acc(v3);
        }
        for (; r < m1; r += st) acc(zp[(size_t)r * m.groups + m.cg]);
    }
    reduce_columns(m, s, q, sum, sumsq);
}

// y = act((z - mean) * inv_std + beta)
__global__ __launch_bounds__(kThreads) void bn_apply_kernel(const float *__restrict__ z, long long n4, int groups,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ inv_std,
                                                            const float *__restrict__ beta, int relu,
                                                            float *__restrict__ y)
{
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long long)gridDim.x * kThreads) {
        const int cg = (int)(i % groups);
        const float4 v = reinterpret_cast<const float4 *>(z)[i];
        const float4 mu = reinterpret_cast<const float4 *>(mean)[cg], is = reinterpret_cast<const float4 *>(inv_std)[cg],
                     be = reinterpret_cast<const float4 *>(beta)[cg];
        // (one fused multiply-add per element, spelled out: the backward kernels recompute exactly this value to rebuild
        // the ReLU mask from z when they are not given y)
        float4 o = make_float4(__fmaf_rn(v.x - mu.x, is.x, be.x), __fmaf_rn(v.y - mu.y, is.y, be.y),
                               __fmaf_rn(v.z - mu.z, is.z, be.z), __fmaf_rn(v.w - mu.w, is.w, be.w));
        if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
        reinterpret_cast<float4 *>(y)[i] = o;
    }
}

// g = dy * (y > 0) (or dy);  sum_g[c] += sum_m g,  sum_gz[c] += sum_m g * zhat,  zhat = (z - mean) * inv_std
__global__ __launch_bounds__(kRThreads) void bn_bwd_reduce_kernel(const float *__restrict__ dy,
                                                                 const float *__restrict__ y,
                                                                 const float *__restrict__ z, long long M, int C,
                                                                 long long rows_per_block,
                                                                 const float *__restrict__ mean,
                                                                 const float *__restrict__ inv_std,
                                                                 const float *__restrict__ beta,
                                                                 double *__restrict__ sum_g, double *__restrict__ sum_gz)
{
    // the ReLU mask: from y when given; else, with beta, recomputed from z as bn_apply_kernel computed y (r06: one of the
    // three 300 MB streams of the pass less); else none
    const RowMap m(C);
    const long long m0 = (long long)blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    if (m.r0 < m.rstride) {
        const float4 mu = reinterpret_cast<const float4 *>(mean)[m.cg], is = reinterpret_cast<const float4 *>(inv_std)[m.cg];
        const float4 be = (!y && beta) ? reinterpret_cast<const float4 *>(beta)[m.cg] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 *gp = reinterpret_cast<const float4 *>(dy), *yp = reinterpret_cast<const float4 *>(y),
                     *zp = reinterpret_cast<const float4 *>(z);
        const bool masked = y || beta;
        auto acc = [&](float4 g, float4 a, float4 v) __attribute__((always_inline)) {
            if (!y && beta)
                a = make_float4(__fmaf_rn(v.x - mu.x, is.x, be.x), __fmaf_rn(v.y - mu.y, is.y, be.y),
                                __fmaf_rn(v.z - mu.z, is.z, be.z), __fmaf_rn(v.w - mu.w, is.w, be.w));
            if (masked) {
                g.x = a.x > 0.f ? g.x : 0.f; g.y = a.y > 0.f ? g.y : 0.f;
                g.z = a.z > 0.f ? g.z : 0.f; g.w = a.w > 0.f ? g.w : 0.f;
            }
            s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
            q.x += g.x * (v.x - mu.x) * is.x; q.y += g.y * (v.y - mu.y) * is.y;
            q.z += g.z * (v.z - mu.z) * is.z; q.w += g.w * (v.w - mu.w) * is.w;
        };
        long long r = m0 + m.r0;
        const long long st = m.rstride;
        for (; r + 3 * st < m1; r += 4 * st) {
            float4 g[4], a[4], v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t o = (size_t)(r + k * st) * m.groups + m.cg;
                g[k] = gp[o];
                a[k] = y ? yp[o] : g[k];
                v[k] = zp[o];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc(g[k], a[k], v[k]);
        }
        for (; r < m1; r += st) {
            const size_t o = (size_t)r * m.groups + m.cg;
            acc(gp[o], y ? yp[o] : gp[o], zp[o]);
        }
    }
    reduce_columns(m, s, q, sum_g, sum_gz);
}

// dz = inv_std * (g - mean_g - zhat * mean_gz)
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                                const float *__restrict__ z, long long n4, int groups,
                                                                const float *__restrict__ mean,
                                                                const float *__restrict__ inv_std,
                                                                const float *__restrict__ mean_g,
                                                                const float *__restrict__ mean_gz,
                                                                const float *__restrict__ beta, float *__restrict__ dz)
{
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long long)gridDim.x * kThreads) {
        const int cg = (int)(i % groups);
        float4 g = reinterpret_cast<const float4 *>(dy)[i];
        const float4 v = reinterpret_cast<const float4 *>(z)[i];
        const float4 mu = reinterpret_cast<const float4 *>(mean)[cg], is = reinterpret_cast<const float4 *>(inv_std)[cg],
                     mg = reinterpret_cast<const float4 *>(mean_g)[cg], mz = reinterpret_cast<const float4 *>(mean_gz)[cg];
        if (y || beta) {
            float4 a;
            if (y) {
                a = reinterpret_cast<const float4 *>(y)[i];
            } else {  // the mask recomputed from z, as bn_apply_kernel computed y
                const float4 be = reinterpret_cast<const float4 *>(beta)[cg];
                a = make_float4(__fmaf_rn(v.x - mu.x, is.x, be.x), __fmaf_rn(v.y - mu.y, is.y, be.y),
                                __fmaf_rn(v.z - mu.z, is.z, be.z), __fmaf_rn(v.w - mu.w, is.w, be.w));
            }
            g.x = a.x > 0.f ? g.x : 0.f; g.y = a.y > 0.f ? g.y : 0.f;
            g.z = a.z > 0.f ? g.z : 0.f; g.w = a.w > 0.f ? g.w : 0.f;
        }
        reinterpret_cast<float4 *>(dz)[i] =
            make_float4(is.x * (g.x - mg.x - (v.x - mu.x) * is.x * mz.x), is.y * (g.y - mg.y - (v.y - mu.y) * is.y * mz.y),
                        is.z * (g.z - mg.z - (v.z - mu.z) * is.z * mz.z), is.w * (g.w - mg.w - (v.w - mu.w) * is.w * mz.w));
    }
}

int check_mc(const char *op, long long M, int C)
{
    MPSR_REQUIRE(M >= 1 && C >= 4 && C % 4 == 0 && C <= 1024, "%s: M=%lld C=%d (C must be a multiple of 4, <= 1024)", op,
                 M, C);
    return MPSR_OK;
}

inline void slicing(long long M, int &blocks, long long &rows)
{
    long long b = (M + 255) / 256;
    if (b > kMaxBlocks) b = kMaxBlocks;
    rows = (M + b - 1) / b;
    blocks = (int)((M + rows - 1) / rows);
}

// zero two fp64 arrays of C: one fill when the caller keeps them back to back (a (2, C) tensor), else two
inline hipError_t zero_pair(double *a, double *b, int C, hipStream_t s)
{
    if (b == a + C) return hipMemsetAsync(a, 0, sizeof(double) * 2 * C, s);
    if (hipError_t e = hipMemsetAsync(a, 0, sizeof(double) * C, s)) return e;
    return hipMemsetAsync(b, 0, sizeof(double) * C, s);
}

inline int stream_grid(long long n4) { return (int)((n4 + kThreads - 1) / kThreads < 65536 ? (n4 + kThreads - 1) / kThreads : 65536); }

// The per-channel arithmetic between the two passes of a layer, as ONE launch (r06: it was ~16 ATen launches of 3-5 us
// on C-sized tensors per layer and step): mean / biased variance from the shifted fp64 sums, the moving statistics'
// update of the fused TensorFlow kernel (unbiased variance into the average), mean and 1 / sqrt(var + eps) as floats.
__global__ void bn_finalize_kernel(const double *__restrict__ sum, const double *__restrict__ sumsq,
                                   const float *__restrict__ z0, double inv_m, double unbias, double eps, float decay,
                                   int C, float *__restrict__ moving_mean, float *__restrict__ moving_var,
                                   float *__restrict__ mean, float *__restrict__ inv_std)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double d = sum[c] * inv_m;
    const double mu = (double)z0[c] + d;
    const double var = fmax(sumsq[c] * inv_m - d * d, 0.0);
    mean[c] = (float)mu;
    inv_std[c] = (float)(1.0 / sqrt(var + eps));
    if (moving_mean) moving_mean[c] = moving_mean[c] * decay + (1.f - decay) * (float)mu;
    if (moving_var) moving_var[c] = moving_var[c] * decay + (1.f - decay) * (float)(var * unbias);
}

// Between the backward's two passes: the beta gradient into its slot of the flat buffer, the two means as floats.
__global__ void bn_grad_finalize_kernel(const double *__restrict__ sum_g, const double *__restrict__ sum_gz,
                                        double inv_count, int C, float *__restrict__ dbeta, float *__restrict__ mean_g,
                                        float *__restrict__ mean_gz)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (dbeta) dbeta[c] += (float)sum_g[c];
    mean_g[c] = (float)(sum_g[c] * inv_count);
    mean_gz[c] = (float)(sum_gz[c] * inv_count);
}


}  // namespace

extern "C" int mpsr_batch_norm_stats(const float *z, long long M, int C, double *sum, double *sumsq_shifted,
                                     mpsr_stream_t stream)
{
    if (int rc = check_mc("batch_norm_stats", M, C)) return rc;
    MPSR_REQUIRE(z && sum && sumsq_shifted && ((uintptr_t)z & 15) == 0, "batch_norm_stats: null or misaligned pointer");
    hipStream_t s = mpsr::as_stream(stream);
    MPSR_CHECK_HIP(zero_pair(sum, sumsq_shifted, C, s));
    int blocks;
    long long rows;
    slicing(M, blocks, rows);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(blocks), dim3(kRThreads), 0, s, z, M, C, rows, sum, sumsq_shifted);
    MPSR_CHECK_LAUNCH("bn_stats_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_batch_norm_apply(const float *z, long long M, int C, const float *mean, const float *inv_std,
                                     const float *beta, int relu, float *y, mpsr_stream_t stream)
{
    if (int rc = check_mc("batch_norm_apply", M, C)) return rc;
    MPSR_REQUIRE(z && mean && inv_std && beta && y, "batch_norm_apply: null pointer");
    const long long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(stream_grid(n4)), dim3(kThreads), 0, mpsr::as_stream(stream), z, n4, C / 4,
                       mean, inv_std, beta, relu, y);
    MPSR_CHECK_LAUNCH("bn_apply_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_batch_norm_grad_sums(const float *dy, const float *y, const float *z, long long M, int C,
                                         const float *mean, const float *inv_std, double *sum_g, double *sum_gz,
                                         mpsr_stream_t stream)
{
    if (int rc = check_mc("batch_norm_grad_sums", M, C)) return rc;
    MPSR_REQUIRE(dy && z && mean && inv_std && sum_g && sum_gz, "batch_norm_grad_sums: null pointer");
    hipStream_t s = mpsr::as_stream(stream);
    MPSR_CHECK_HIP(zero_pair(sum_g, sum_gz, C, s));
    int blocks;
    long long rows;
    slicing(M, blocks, rows);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(blocks), dim3(kRThreads), 0, s, dy, y, z, M, C, rows, mean, inv_std,
                       (const float *)nullptr, sum_g, sum_gz);
    MPSR_CHECK_LAUNCH("bn_bwd_reduce_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_batch_norm_grad(const float *dy, const float *y, const float *z, long long M, int C,
                                    const float *mean, const float *inv_std, const float *mean_g, const float *mean_gz,
                                    float *dz, mpsr_stream_t stream)
{
    if (int rc = check_mc("batch_norm_grad", M, C)) return rc;
    MPSR_REQUIRE(dy && z && mean && inv_std && mean_g && mean_gz && dz, "batch_norm_grad: null pointer");
    const long long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(n4)), dim3(kThreads), 0, mpsr::as_stream(stream), dy, y, z,
                       n4, C / 4, mean, inv_std, mean_g, mean_gz, (const float *)nullptr, dz);
    MPSR_CHECK_LAUNCH("bn_bwd_apply_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_batch_norm_finalize(const double *sum, const double *sumsq_shifted, const float *z_row0, long long M,
                                        int C, float eps, float decay, float *moving_mean, float *moving_variance,
                                        float *mean, float *inv_std, mpsr_stream_t stream)
{
    if (int rc = check_mc("batch_norm_finalize", M, C)) return rc;
    MPSR_REQUIRE(sum && sumsq_shifted && z_row0 && mean && inv_std, "batch_norm_finalize: null pointer");
    MPSR_REQUIRE(eps >= 0.f && decay >= 0.f && decay <= 1.f, "batch_norm_finalize: eps >= 0 and 0 <= decay <= 1");
    const double unbias = (double)M / (double)(M > 1 ? M - 1 : 1);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, mpsr::as_stream(stream), sum,
                       sumsq_shifted, z_row0, 1.0 / (double)M, unbias, (double)eps, decay, C, moving_mean, moving_variance,
                       mean, inv_std);
    MPSR_CHECK_LAUNCH("bn_finalize_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_batch_norm_grad_finalize(const double *sum_g, const double *sum_gz, double count, int C, float *dbeta,
                                             float *mean_g, float *mean_gz, mpsr_stream_t stream)
{
    MPSR_REQUIRE(C > 0 && count >= 1.0, "batch_norm_grad_finalize: C > 0 and count >= 1");
    MPSR_REQUIRE(sum_g && sum_gz && mean_g && mean_gz, "batch_norm_grad_finalize: null pointer");
    hipLaunchKernelGGL(bn_grad_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, mpsr::as_stream(stream), sum_g, sum_gz,
                       1.0 / count, C, dbeta, mean_g, mean_gz);
    MPSR_CHECK_LAUNCH("bn_grad_finalize_kernel");
    return MPSR_OK;
}

// The two backward passes with the ReLU mask REBUILT from z (y is not read: relu((z - mean) * inv_std + beta) > 0 is
// recomputed with bn_apply_kernel's own fused multiply-add, so the mask is the one y > 0 gives, bit for bit).
extern "C" int mpsr_batch_norm_grad_sums_z(const float *dy, const float *z, long long M, int C, const float *mean,
                                           const float *inv_std, const float *beta, double *sum_g, double *sum_gz,
                                           mpsr_stream_t stream)
{
    if (int rc = check_mc("batch_norm_grad_sums_z", M, C)) return rc;
    MPSR_REQUIRE(dy && z && mean && inv_std && beta && sum_g && sum_gz, "batch_norm_grad_sums_z: null pointer");
    hipStream_t s = mpsr::as_stream(stream);
    MPSR_CHECK_HIP(zero_pair(sum_g, sum_gz, C, s));
    int blocks;
    long long rows;
    slicing(M, blocks, rows);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(blocks), dim3(kRThreads), 0, s, dy, (const float *)nullptr, z, M, C, rows,
                       mean, inv_std, beta, sum_g, sum_gz);
    MPSR_CHECK_LAUNCH("bn_bwd_reduce_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_batch_norm_grad_z(const float *dy, const float *z, long long M, int C, const float *mean,
                                      const float *inv_std, const float *beta, const float *mean_g, const float *mean_gz,
                                      float *dz, mpsr_stream_t stream)
{
    if (int rc = check_mc("batch_norm_grad_z", M, C)) return rc;
    MPSR_REQUIRE(dy && z && mean && inv_std && beta && mean_g && mean_gz && dz, "batch_norm_grad_z: null pointer");
    const long long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(n4)), dim3(kThreads), 0, mpsr::as_stream(stream), dy,
                       (const float *)nullptr, z, n4, C / 4, mean, inv_std, mean_g, mean_gz, beta, dz);
    MPSR_CHECK_LAUNCH("bn_bwd_apply_kernel");
    return MPSR_OK;
}
