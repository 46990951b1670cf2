// Per-variable gradient clipping over a flat gradient buffer: slim.learning.create_train_op(clip_gradient_norm=1.0)
// applies tf.clip_by_norm to every variable's gradient separately (reference core/trainer.py:78-81).  With ~220
// variables that is ~1000 tiny launches when done tensor by tensor; here it is two launches over a chunk table
// (one workgroup per <= 16 Ki-float chunk of one variable): squared sums -> per-variable scale.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void seg_sumsq_kernel(const float *__restrict__ g,
                                                             const int *__restrict__ chunk_seg,
                                                             const long long *__restrict__ chunk_begin,
                                                             const int *__restrict__ chunk_len,
                                                             float *__restrict__ sumsq)
{
    __shared__ float scratch[kThreads / 64];
    const int c = blockIdx.x;
    const float *p = g + chunk_begin[c];
    const int n = chunk_len[c];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += kThreads) s += p[i] * p[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(&sumsq[chunk_seg[c]], scratch[0] + scratch[1] + scratch[2] + scratch[3]);
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

}  // namespace

extern "C" int mpsr_clip_by_norm_segments(float *grads, const int *chunk_seg, const long long *chunk_begin,
                                          const int *chunk_len, int n_chunks, float *sumsq, int n_segments,
                                          float clip_norm, mpsr_stream_t stream)
{
    MPSR_REQUIRE(n_chunks >= 0 && n_segments >= 0 && clip_norm > 0.f, "clip_by_norm_segments: bad arguments");
    if (n_chunks == 0 || n_segments == 0) return MPSR_OK;
    MPSR_REQUIRE(grads && chunk_seg && chunk_begin && chunk_len && sumsq, "clip_by_norm_segments: null pointer");
    hipStream_t s = mpsr::as_stream(stream);
    MPSR_CHECK_HIP(hipMemsetAsync(sumsq, 0, sizeof(float) * (size_t)n_segments, s));
    hipLaunchKernelGGL(seg_sumsq_kernel, dim3(n_chunks), dim3(kThreads), 0, s, grads, chunk_seg, chunk_begin, chunk_len,
                       sumsq);
    MPSR_CHECK_LAUNCH("seg_sumsq_kernel");
    hipLaunchKernelGGL(seg_scale_kernel, dim3(n_chunks), dim3(kThreads), 0, s, grads, chunk_seg, chunk_begin, chunk_len,
                       sumsq, clip_norm);
    MPSR_CHECK_LAUNCH("seg_scale_kernel");
    return MPSR_OK;
}
