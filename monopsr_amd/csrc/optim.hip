// Per-variable gradient clipping over a flat gradient buffer: slim.learning.create_train_op(clip_gradient_norm=1.0)
// applies tf.clip_by_norm to every variable's gradient separately (reference core/trainer.py:78-81).  With ~220
// variables that is ~1000 tiny launches when done tensor by tensor; here it is two launches over a chunk table
// (one workgroup per <= 16 Ki-float chunk of one variable): squared sums -> per-variable scale (r06: three, the sums
// are reduced in a fixed order instead of with atomics).
#include <cmath>

#include "common.h"

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void seg_sumsq_kernel(const float *__restrict__ g,
                                                             const int *__restrict__ chunk_seg,
                                                             const long long *__restrict__ chunk_begin,
                                                             const int *__restrict__ chunk_len,
                                                             float *__restrict__ partial)
{
    __shared__ float scratch[kThreads / 64];
    const int c = blockIdx.x;
    const float *p = g + chunk_begin[c];
    const int n = chunk_len[c];
    float s = 0.f;
    if ((((uintptr_t)p) & 15) == 0) {  // 16-byte loads (every tensor of the flat buffer is 256-byte aligned, chunks 64 KiB)
        const float4 *p4 = reinterpret_cast<const float4 *>(p);
        const int n4 = n >> 2;
        for (int i = threadIdx.x; i < n4; i += kThreads) {
            const float4 v = p4[i];
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (int i = (n4 << 2) + threadIdx.x; i < n; i += kThreads) s += p[i] * p[i];
    } else {
        for (int i = threadIdx.x; i < n; i += kThreads) s += p[i] * p[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = s;
    __syncthreads();
    // one partial per chunk, no atomics: the variable's norm is summed from them in a FIXED order (seg_sum_kernel), so
    // that every replica of a data-parallel run scales the same reduced gradient by the same bits (r06: with an atomic
    // accumulation two ranks holding bit-identical reduced gradients updated their parameters one ulp apart --
    // tests/test_sharded_training_step_gpu.py)
    if (threadIdx.x == 0) partial[c] = scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// sumsq[s] = sum of the partials of variable s's chunks (consecutive in the table: chunk_seg ascends), in a fixed order.
// One wave per variable.
__global__ __launch_bounds__(64) void seg_sum_kernel(const float *__restrict__ partial, const int *__restrict__ chunk_seg,
                                                     int n_chunks, float *__restrict__ sumsq)
{
    const int s = blockIdx.x;
    int lo = 0, hi = n_chunks;  // first chunk with chunk_seg >= s
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (chunk_seg[mid] < s) lo = mid + 1;
        else hi = mid;
    }
    const int first = lo;
    hi = n_chunks;  // first chunk with chunk_seg > s
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (chunk_seg[mid] <= s) lo = mid + 1;
        else hi = mid;
    }
    float v = 0.f;
    for (int c = first + (int)threadIdx.x; c < lo; c += 64) v += partial[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (threadIdx.x == 0) sumsq[s] = v;
}

// tf.clip_by_norm: g * clip / max(||g||, clip)
__global__ __launch_bounds__(kThreads) void seg_scale_kernel(float *__restrict__ g, const int *__restrict__ chunk_seg,
                                                             const long long *__restrict__ chunk_begin,
                                                             const int *__restrict__ chunk_len,
                                                             const float *__restrict__ sumsq, float clip)
{
    const int c = blockIdx.x;
    const float norm = sqrtf(sumsq[chunk_seg[c]]);
    if (!(norm > clip)) return;  // also leaves NaN norms untouched, as a multiply by NaN would not help either
    const float scale = clip / norm;
    float *p = g + chunk_begin[c];
    const int n = chunk_len[c];
    for (int i = threadIdx.x; i < n; i += kThreads) p[i] *= scale;
}

// clip -> Adam -> moving average of ONE chunk of one variable, in one pass over the flat buffers (r06): the per-variable
// scale is applied to the gradient on its way into the update instead of being written back (seg_scale_kernel), and the
// moving average is updated from the parameter just computed instead of re-reading it (a torch lerp_): 9 streams of the
// flat buffer where the three launches moved 13.  Same arithmetic, operation by operation, as seg_scale_kernel ->
// adam_kernel (backward.hip) -> shadow += (1 - decay) * (p - shadow).
__global__ __launch_bounds__(kThreads) void clip_adam_ema_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                                 float *__restrict__ m, float *__restrict__ v,
                                                                 float *__restrict__ shadow,
                                                                 const int *__restrict__ chunk_seg,
                                                                 const long long *__restrict__ chunk_begin,
                                                                 const int *__restrict__ chunk_len,
                                                                 const float *__restrict__ sumsq, float clip, float lr_t,
                                                                 float b1, float b2, float eps, float ema_w)
{
    const int c = blockIdx.x;
    float scale = 1.f;
    bool scaled = false;
    if (sumsq) {
        const float norm = sqrtf(sumsq[chunk_seg[c]]);
        scaled = norm > clip;  // (a NaN norm leaves the gradient as it is, like seg_scale_kernel)
        if (scaled) scale = clip / norm;
    }
    const long long o = chunk_begin[c];
    const int n = chunk_len[c];
    for (int i = threadIdx.x; i < n; i += kThreads) {
        const float gr = g[o + i];
        const float gi = scaled ? gr * scale : gr;
        const float mi = b1 * m[o + i] + (1.f - b1) * gi;
        const float vi = b2 * v[o + i] + (1.f - b2) * gi * gi;
        m[o + i] = mi;
        v[o + i] = vi;
        const float pi = p[o + i] - lr_t * mi / (sqrtf(vi) + eps);
        p[o + i] = pi;
        if (shadow) {
            const float sh = shadow[o + i];
            shadow[o + i] = sh + ema_w * (pi - sh);
        }
    }
}

}  // namespace

// squared norm of every variable into sumsq[0 .. n_segments), deterministically; sumsq must hold n_segments + n_chunks
// floats (the chunk partials live behind the norms)
static int segment_norms(const float *grad, const int *chunk_seg, const long long *chunk_begin, const int *chunk_len,
                         int n_chunks, float *sumsq, size_t sumsq_floats, int n_segments, hipStream_t s, const char *op)
{
    if (sumsq_floats < (size_t)n_segments + (size_t)n_chunks)
        return mpsr::fail(MPSR_ERR_WORKSPACE, "%s: sumsq holds %zu floats, needs n_segments + n_chunks = %zu", op,
                          sumsq_floats, (size_t)n_segments + (size_t)n_chunks);
    float *partial = sumsq + n_segments;
    hipLaunchKernelGGL(seg_sumsq_kernel, dim3(n_chunks), dim3(kThreads), 0, s, grad, chunk_seg, chunk_begin, chunk_len,
                       partial);
    MPSR_CHECK_LAUNCH("seg_sumsq_kernel");
    hipLaunchKernelGGL(seg_sum_kernel, dim3(n_segments), dim3(64), 0, s, (const float *)partial, chunk_seg, n_chunks, sumsq);
    MPSR_CHECK_LAUNCH("seg_sum_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_clip_adam_ema_step(float *param, const float *grad, float *m, float *v, float *shadow,
                                       const int *chunk_seg, const long long *chunk_begin, const int *chunk_len,
                                       int n_chunks, float *sumsq, size_t sumsq_floats, int n_segments, float clip_norm,
                                       float lr, float beta1, float beta2, float eps, int step, float ema_decay,
                                       mpsr_stream_t stream)
{
    MPSR_REQUIRE(n_chunks >= 0 && n_segments >= 0 && step >= 1, "clip_adam_ema_step: bad arguments");
    if (n_chunks == 0) return MPSR_OK;
    MPSR_REQUIRE(param && grad && m && v && chunk_seg && chunk_begin && chunk_len, "clip_adam_ema_step: null pointer");
    MPSR_REQUIRE(!(clip_norm > 0.f) || (sumsq && n_segments > 0), "clip_adam_ema_step: clipping needs the sumsq scratch");
    hipStream_t s = mpsr::as_stream(stream);
    const bool clip = clip_norm > 0.f;
    if (clip)
        if (int rc = segment_norms(grad, chunk_seg, chunk_begin, chunk_len, n_chunks, sumsq, sumsq_floats, n_segments, s,
                                   "clip_adam_ema_step"))
            return rc;
    const double c1 = 1.0 - pow((double)beta1, step), c2 = 1.0 - pow((double)beta2, step);
    const float lr_t = (float)(lr * sqrt(c2) / c1);  // (as mpsr_adam_step)
    hipLaunchKernelGGL(clip_adam_ema_kernel, dim3(n_chunks), dim3(kThreads), 0, s, param, grad, m, v, shadow, chunk_seg,
                       chunk_begin, chunk_len, clip ? (const float *)sumsq : (const float *)nullptr, clip_norm, lr_t, beta1,
                       beta2, eps, 1.f - ema_decay);
    MPSR_CHECK_LAUNCH("clip_adam_ema_kernel");
    return MPSR_OK;
}

extern "C" int mpsr_clip_by_norm_segments(float *grads, const int *chunk_seg, const long long *chunk_begin,
                                          const int *chunk_len, int n_chunks, float *sumsq, size_t sumsq_floats,
                                          int n_segments, float clip_norm, mpsr_stream_t stream)
{
    MPSR_REQUIRE(n_chunks >= 0 && n_segments >= 0 && clip_norm > 0.f, "clip_by_norm_segments: bad arguments");
    if (n_chunks == 0 || n_segments == 0) return MPSR_OK;
    MPSR_REQUIRE(grads && chunk_seg && chunk_begin && chunk_len && sumsq, "clip_by_norm_segments: null pointer");
    hipStream_t s = mpsr::as_stream(stream);
    if (int rc = segment_norms(grads, chunk_seg, chunk_begin, chunk_len, n_chunks, sumsq, sumsq_floats, n_segments, s,
                               "clip_by_norm_segments"))
        return rc;
    hipLaunchKernelGGL(seg_scale_kernel, dim3(n_chunks), dim3(kThreads), 0, s, grads, chunk_seg, chunk_begin, chunk_len,
                       sumsq, clip_norm);
    MPSR_CHECK_LAUNCH("seg_scale_kernel");
    return MPSR_OK;
}
