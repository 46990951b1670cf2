// Gradient of tf.image.crop_and_resize w.r.t. the image (TF's CropAndResizeGradImage): every crop sample that fell
// inside the image hands its gradient to the four pixels it interpolated, with the bilinear weights of the forward
// pass (image_ops.hip: same float32 coordinate arithmetic, no contraction).  Needed once the full-image trunk is
// trained: its 40x152x1024 feature map is cropped per proposal (net_builder.py:54-59) and must receive the boxes'
// gradients.  Scatter with fp32 atomics into the pre-zeroed image gradient.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"

#pragma clang fp contract(off)

namespace {

__global__ __launch_bounds__(256) void crop_and_resize_grad_kernel(const float *__restrict__ dy, int H, int W, int C,
                                                                   const float *__restrict__ boxes,
                                                                   const int *__restrict__ box_ind, int nimg, int ch,
                                                                   int cw, float *__restrict__ dimg, long long total)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int x = (int)(r % cw);
        r /= cw;
        const int y = (int)(r % ch);
        const int bi = (int)(r / ch);
        const float y1 = boxes[4 * bi], x1 = boxes[4 * bi + 1], y2 = boxes[4 * bi + 2], x2 = boxes[4 * bi + 3];
        const int img = box_ind ? box_ind[bi] : 0;
        const float hs = ch > 1 ? (y2 - y1) * (float)(H - 1) / (float)(ch - 1) : 0.f;
        const float ws = cw > 1 ? (x2 - x1) * (float)(W - 1) / (float)(cw - 1) : 0.f;
        const float in_y = ch > 1 ? y1 * (float)(H - 1) + (float)y * hs : 0.5f * (y1 + y2) * (float)(H - 1);
        const float in_x = cw > 1 ? x1 * (float)(W - 1) + (float)x * ws : 0.5f * (x1 + x2) * (float)(W - 1);
        if (!(img >= 0 && img < nimg && !(in_y < 0.f) && !(in_y > (float)(H - 1)) && !(in_x < 0.f) &&
              !(in_x > (float)(W - 1)) && in_y == in_y && in_x == in_x))
            continue;  // extrapolated samples carry no image gradient
        const int top = (int)floorf(in_y), bot = (int)ceilf(in_y);
        const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
        const float yl = in_y - (float)top, xl = in_x - (float)left;
        const float g = dy[i];
        float *base = dimg + (size_t)img * H * W * C + c;
        unsafeAtomicAdd(base + ((size_t)top * W + left) * C, g * (1.f - yl) * (1.f - xl));
        unsafeAtomicAdd(base + ((size_t)top * W + right) * C, g * (1.f - yl) * xl);
        unsafeAtomicAdd(base + ((size_t)bot * W + left) * C, g * yl * (1.f - xl));
        unsafeAtomicAdd(base + ((size_t)bot * W + right) * C, g * yl * xl);
    }
}

}  // namespace

extern "C" int mpsr_crop_and_resize_grad(const float *grad_out, int nimg, int H, int W, int C, const float *boxes,
                                         const int *box_ind, int nb, int ch, int cw, float *grad_image,
                                         mpsr_stream_t stream)
{
    MPSR_REQUIRE(nimg > 0 && H > 0 && W > 0 && C > 0 && nb >= 0 && ch > 0 && cw > 0,
                 "crop_and_resize_grad: bad shape (nimg=%d H=%d W=%d C=%d nb=%d crop=%dx%d)", nimg, H, W, C, nb, ch,
                 cw);
    MPSR_REQUIRE(grad_image, "crop_and_resize_grad: null pointer");
    hipStream_t s = mpsr::as_stream(stream);
    MPSR_CHECK_HIP(hipMemsetAsync(grad_image, 0, sizeof(float) * (size_t)nimg * H * W * C, s));
    if (nb == 0) return MPSR_OK;
    MPSR_REQUIRE(grad_out && boxes, "crop_and_resize_grad: null pointer");
    const long long total = (long long)nb * ch * cw * C;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(crop_and_resize_grad_kernel, dim3((unsigned)(blocks < 262144 ? blocks : 262144)), dim3(256), 0, s,
                       grad_out, H, W, C, boxes, box_ind, nimg, ch, cw, grad_image, total);
    MPSR_CHECK_LAUNCH("crop_and_resize_grad_kernel");
    return MPSR_OK;
}
