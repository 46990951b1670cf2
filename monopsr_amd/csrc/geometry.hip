// Per-box geometry after the heads and the map-shaped loss terms (row 3):
//   local -> global instance xyz maps, projection-error maps, global depth maps (+ their gradients), the masked
//   smooth-L1 ("huber") sums the training loss takes over maps, and the post-processing that turns head outputs
//   into KITTI boxes.  All of it is HBM-bound elementwise / per-instance-reduction work: one pass over each map,
//   coalesced, reductions done by ONE workgroup per instance in a fixed order (deterministic, no atomics).
// Reference: datasets/kitti/instance_utils.py:567-680,738-788,988-1032; monopsr_output_builder.py:663-772,805-860;
// monopsr_model.py:960-1071; losses_custom.py:93-132.
#include "common.h"

namespace {

constexpr int kThreads = 256;

// Sum over the workgroup in a fixed tree order; result valid in thread 0 (and returned to all via LDS).
__device__ __forceinline__ float block_sum(float v, float *scratch)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < kThreads / 64; ++i) r += scratch[i];
    return r;
}

// --------------------------------------------------------------------------------- local -> global xyz maps

// instance_utils.py:567-602: p' = T(centroid) * R_y(view) * p  (transform_utils.py:92-104).
__global__ __launch_bounds__(kThreads) void local_to_global_kernel(const float *__restrict__ xyz,
                                                                   const float *__restrict__ view,
                                                                   const float *__restrict__ cen,
                                                                   float *__restrict__ out, int P)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= P) return;
    const float c = cosf(view[b]), s = sinf(view[b]);
    const float *p = xyz + ((size_t)b * P + i) * 3;
    const float x = p[0], y = p[1], z = p[2];
    float *o = out + ((size_t)b * P + i) * 3;
    o[0] = (c * x + s * z) + cen[b * 3 + 0];
    o[1] = y + cen[b * 3 + 1];
    o[2] = (-s * x + c * z) + cen[b * 3 + 2];
}

// grad_local = R^T g ; grad_centroid[b] = sum_p g.
__global__ __launch_bounds__(kThreads) void local_to_global_grad_kernel(const float *__restrict__ g,
                                                                        const float *__restrict__ view,
                                                                        float *__restrict__ grad_local,
                                                                        float *__restrict__ grad_cen, int P)
{
    __shared__ float scratch[kThreads / 64];
    const int b = blockIdx.x;
    const float c = cosf(view[b]), s = sinf(view[b]);
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int i = threadIdx.x; i < P; i += kThreads) {
        const float *q = g + ((size_t)b * P + i) * 3;
        const float gx = q[0], gy = q[1], gz = q[2];
        if (grad_local) {
            float *o = grad_local + ((size_t)b * P + i) * 3;
            o[0] = c * gx - s * gz;
            o[1] = gy;
            o[2] = s * gx + c * gz;
        }
        sx += gx;
        sy += gy;
        sz += gz;
    }
    if (grad_cen) {
        sx = block_sum(sx, scratch);
        sy = block_sum(sy, scratch);
        sz = block_sum(sz, scratch);
        if (threadIdx.x == 0) {
            grad_cen[b * 3 + 0] = sx;
            grad_cen[b * 3 + 1] = sy;
            grad_cen[b * 3 + 2] = sz;
        }
    }
}

// ------------------------------------------------------------------------------------- projection error

struct BoxGrid {  // tf.linspace start/step of the expected pixel-centre grid (instance_utils.py:738-788)
    float u0, du, v0, dv, bw, bh;
};

__device__ __forceinline__ BoxGrid box_grid(const float *box, int H, int W)
{
    const float v1 = box[0], u1 = box[1], v2 = box[2], u2 = box[3];
    const float hu = (u2 - u1) / (float)W / 2.0f, hv = (v2 - v1) / (float)H / 2.0f;
    BoxGrid g;
    g.u0 = u1 + hu;
    g.du = W > 1 ? ((u2 - hu) - g.u0) / (float)(W - 1) : 0.f;
    g.v0 = v1 + hv;
    g.dv = H > 1 ? ((v2 - hv) - g.v0) / (float)(H - 1) : 0.f;
    g.bw = u2 - u1;
    g.bh = v2 - v1;
    return g;
}

// monopsr_output_builder.py:681-746.  One workgroup per instance.  MODE 0: forward (maps optional, norm);
// MODE 1: gradient of norm w.r.t. xyz_global given d(loss)/d(norm).
template <int MODE>
__global__ __launch_bounds__(kThreads) void proj_err_kernel(const float *__restrict__ xyz,
                                                            const float *__restrict__ boxes,
                                                            const float *__restrict__ cam_p,
                                                            const float *__restrict__ mask,
                                                            const float *__restrict__ gnorm,
                                                            float *__restrict__ maps, float *__restrict__ out,
                                                            int H, int W)
{
    __shared__ float scratch[kThreads / 64];
    const int b = blockIdx.x, P = H * W;
    const BoxGrid g = box_grid(boxes + b * 4, H, W);
    float pm[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) pm[i] = cam_p[i];
    float nv = 0.f;
    for (int i = threadIdx.x; i < P; i += kThreads) nv += mask[(size_t)b * P + i];
    nv = block_sum(nv, scratch);
    if (nv < 1.0f) nv = 1.0f;
    const float gscale = MODE == 1 ? gnorm[b] / nv : 0.f;
    float sum = 0.f;
    for (int i = threadIdx.x; i < P; i += kThreads) {
        const int r = i / W, cidx = i - r * W;
        const float *p = xyz + ((size_t)b * P + i) * 3;
        const float x = p[0], y = p[1], z = p[2];
        const float U = pm[0] * x + pm[1] * y + pm[2] * z + pm[3];
        const float V = pm[4] * x + pm[5] * y + pm[6] * z + pm[7];
        const float Wc = pm[8] * x + pm[9] * y + pm[10] * z + pm[11];
        const float u = U / Wc, v = V / Wc;
        const float m = mask[(size_t)b * P + i];
        const float eu_raw = ((g.u0 + g.du * (float)cidx) - u) / g.bw * m;
        const float ev_raw = ((g.v0 + g.dv * (float)r) - v) / g.bh * m;
        if (MODE == 0) {
            const float eu = fminf(fmaxf(eu_raw, -2.0f), 2.0f), ev = fminf(fmaxf(ev_raw, -2.0f), 2.0f);
            if (maps) {
                maps[((size_t)b * P + i) * 2 + 0] = eu;
                maps[((size_t)b * P + i) * 2 + 1] = ev;
            }
            sum += eu + ev;
        } else {
            // d e_u / d u = -m / bw inside the clip range; u = U / Wc
            const float ku = (eu_raw >= -2.0f && eu_raw <= 2.0f) ? -m / g.bw * gscale / Wc : 0.f;
            const float kv = (ev_raw >= -2.0f && ev_raw <= 2.0f) ? -m / g.bh * gscale / Wc : 0.f;
            float *o = out + ((size_t)b * P + i) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) o[k] = ku * (pm[k] - u * pm[8 + k]) + kv * (pm[4 + k] - v * pm[8 + k]);
        }
    }
    if (MODE == 0) {
        sum = block_sum(sum, scratch);
        if (threadIdx.x == 0) out[b] = sum / nv;
    }
}

// ------------------------------------------------------------------------------------ global depth maps

// instance_utils.py:605-680: coefficients a, k such that row r's offset = z * (a + k r)  (the offset is linear in
// the centroid depth z: -(z / cos va) / cos(th - va) * sin(th - va) * sin va).
__device__ __forceinline__ void depth_row_coef(const float *box, float va, const float *cam_p, int H, float &a, float &k)
{
    const float cu = cam_p[2], f = cam_p[0];
    float x1 = box[1], x2 = box[3];
    const float gs = (x2 - x1) / (float)H / 2.0f;
    x1 += gs;
    x2 -= gs;
    const float tl = atan2f((x1 - cu) / f, 1.0f), tr = atan2f((x2 - cu) / f, 1.0f);
    const float inv = 1.0f / cosf(va), sv = sinf(va);
    const float ol = -(inv / cosf(tl - va) * sinf(tl - va) * sv);
    const float orr = -(inv / cosf(tr - va) * sinf(tr - va) * sv);
    a = ol;
    k = H > 1 ? (orr - ol) / (float)(H - 1) : 0.f;
}

__global__ __launch_bounds__(kThreads) void depth_global_kernel(const float *__restrict__ d, int stride,
                                                                const float *__restrict__ cen_z,
                                                                const float *__restrict__ boxes,
                                                                const float *__restrict__ view,
                                                                const float *__restrict__ cam_p,
                                                                float *__restrict__ out, int H, int W, int rotate)
{
    const int b = blockIdx.y, P = H * W;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= P) return;
    const float z = cen_z[b];
    float off = 0.f;
    if (rotate) {
        float a, k;
        depth_row_coef(boxes + b * 4, view[b], cam_p, H, a, k);
        const int r = i / W;
        off = z * a + (z * k) * (float)r;
    }
    out[(size_t)b * P + i] = d[((size_t)b * P + i) * stride] + z + off;
}

// grad_cen_z[b] = sum_{r,c} g[b,r,c] * (1 + a + k r); the gradient w.r.t. the local depth is g itself.
__global__ __launch_bounds__(kThreads) void depth_global_grad_kernel(const float *__restrict__ g,
                                                                     const float *__restrict__ boxes,
                                                                     const float *__restrict__ view,
                                                                     const float *__restrict__ cam_p,
                                                                     float *__restrict__ grad_cen_z, int H, int W,
                                                                     int rotate)
{
    __shared__ float scratch[kThreads / 64];
    const int b = blockIdx.x, P = H * W;
    float a = 0.f, k = 0.f;
    if (rotate) depth_row_coef(boxes + b * 4, view[b], cam_p, H, a, k);
    float s = 0.f;
    for (int i = threadIdx.x; i < P; i += kThreads) s += g[(size_t)b * P + i] * (1.0f + a + k * (float)(i / W));
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) grad_cen_z[b] = s;
}

// ------------------------------------------------------------------------------------ masked smooth-L1 sums

// tf.losses.huber_loss terms (0.5 q^2 + delta (|e| - q), q = min(|e|, delta)) times a per-pixel weight broadcast
// over C channels.  sums[b] = sum over the instance; counts[b] = C * (number of non-zero weights)
// (Reduction.SUM_BY_NONZERO_WEIGHTS counts the broadcast weights, losses_custom.py:125-132).
__global__ __launch_bounds__(kThreads) void huber_sum_kernel(const float *__restrict__ pred,
                                                             const float *__restrict__ target,
                                                             const float *__restrict__ weights, int P, int C,
                                                             float delta, float *__restrict__ sums,
                                                             float *__restrict__ counts)
{
    __shared__ float scratch[kThreads / 64];
    const int b = blockIdx.x;
    float s = 0.f, n = 0.f;
    for (int i = threadIdx.x; i < P * C; i += kThreads) {
        const float w = weights[(size_t)b * P + i / C];
        const float e = fabsf(pred[(size_t)b * P * C + i] - target[(size_t)b * P * C + i]);
        const float q = fminf(e, delta);
        s += (0.5f * q * q + delta * (e - q)) * w;
        n += w != 0.f ? 1.0f : 0.f;
    }
    s = block_sum(s, scratch);
    n = block_sum(n, scratch);
    if (threadIdx.x == 0) {
        sums[b] = s;
        counts[b] = n;
    }
}

// grad = scale[0] * w * clamp(pred - target, -delta, delta)
__global__ __launch_bounds__(kThreads) void huber_grad_kernel(const float *__restrict__ pred,
                                                              const float *__restrict__ target,
                                                              const float *__restrict__ weights,
                                                              const float *__restrict__ scale, long long n, int C,
                                                              float delta, float *__restrict__ grad)
{
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const float e = pred[i] - target[i];
    grad[i] = scale[0] * weights[i / C] * fminf(fmaxf(e, -delta), delta);
}

// --------------------------------------------------------------------------------------- post-processing

struct FormatParams {
    const float *lwh, *view, *bins, *regs, *cen, *boxes, *scores, *cam_p;
    const int *cls;
    float *box3d, *box2d;
    int B, nb, img_h, img_w, centroid_middle, post_cen_x;
    float max_depth;
};

// obj_utils.py:835-864 corners of [x,y,z,l,w,h,ry] projected with P (calib_utils.py:245-260): min/max u, v.
__device__ void project_corners(const double *b3, const double *pm, double &umin, double &vmin, double &umax,
                                double &vmax)
{
    const double c = cos(b3[6]), s = sin(b3[6]);
    const double hl = b3[3] / 2, hw = b3[4] / 2, h = b3[5];
    const double xs[8] = {hl, hl, -hl, -hl, hl, hl, -hl, -hl};
    const double ys[8] = {0, 0, 0, 0, -h, -h, -h, -h};
    const double zs[8] = {hw, -hw, -hw, hw, hw, -hw, -hw, hw};
    umin = vmin = 1e300;
    umax = vmax = -1e300;
    for (int i = 0; i < 8; ++i) {
        const double x = c * xs[i] + s * zs[i] + b3[0], y = ys[i] + b3[1], z = -s * xs[i] + c * zs[i] + b3[2];
        const double U = pm[0] * x + pm[1] * y + pm[2] * z + pm[3], V = pm[4] * x + pm[5] * y + pm[6] * z + pm[7],
                     Wc = pm[8] * x + pm[9] * y + pm[10] * z + pm[11];
        const double u = U / Wc, v = V / Wc;
        umin = fmin(umin, u);
        umax = fmax(umax, u);
        vmin = fmin(vmin, v);
        vmax = fmax(vmax, v);
    }
}

// monopsr_model.py:960-1071 (test mode, alpha 'dc') + instance_utils.py:988-1032 + monopsr_output_builder.py:805-860.
// One thread per box, fp64 like the reference's numpy.
__global__ void format_boxes_kernel(const FormatParams p)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.B) return;
    double pm[12];
    for (int k = 0; k < 12; ++k) pm[k] = p.cam_p[k];
    double b3[7];
    b3[3] = p.lwh[i * 3 + 0];
    b3[4] = p.lwh[i * 3 + 1];
    b3[5] = p.lwh[i * 3 + 2];
    int best = 0;
    for (int k = 1; k < p.nb; ++k)
        if (p.bins[i * p.nb + k] > p.bins[i * p.nb + best]) best = k;  // np.argmax: first maximum
    const double two_pi = 2.0 * M_PI;
    double alpha = best * (two_pi / p.nb) + (double)p.regs[i * p.nb + best];  // orientation_encoder.py:83-107
    if (alpha < -M_PI) alpha += two_pi;
    if (alpha > M_PI) alpha -= two_pi;
    b3[6] = alpha + (double)p.view[i];
    b3[0] = p.cen[i * 3 + 0];
    b3[1] = (double)p.cen[i * 3 + 1] + (p.centroid_middle ? b3[5] / 2 : 0.0);
    b3[2] = p.cen[i * 3 + 2];
    const double by1 = p.boxes[i * 4 + 0], bx1 = p.boxes[i * 4 + 1], by2 = p.boxes[i * 4 + 2], bx2 = p.boxes[i * 4 + 3];
    double umin, vmin, umax, vmax;
    if (p.post_cen_x) {
        project_corners(b3, pm, umin, vmin, umax, vmax);
        const double cu = (pm[0] * b3[0] + pm[1] * b3[1] + pm[2] * b3[2] + pm[3]) /
                          (pm[8] * b3[0] + pm[9] * b3[1] + pm[10] * b3[2] + pm[11]);
        const double ratio = (cu - umin) / (umax - umin);
        const double u = bx1 + ratio * (bx2 - bx1);
        b3[0] = (u - pm[2]) * (b3[2] / pm[0]);
    }
    // score_boxes: box fit of the (truncated) projection against the detection, and depth
    project_corners(b3, pm, umin, vmin, umax, vmax);
    const double iw = p.img_w, ih = p.img_h;
    double fit;
    if (umin > iw || vmin > ih || umax < 0 || vmax < 0 || (umax - umin) > iw * 0.8 || (vmax - vmin) > ih * 0.8) {
        fit = 0.1;  // box_3d_projector.py:57-71 returns None
    } else {
        const double x1 = fmax(umin, 0.0), y1 = fmax(vmin, 0.0), x2 = fmin(umax, iw), y2 = fmin(vmax, ih);
        const double hh = by2 - by1, ww = bx2 - bx1;
        fit = 1.0 - (fabs((bx1 - x1) / ww) + fabs((bx2 - x2) / ww) + fabs((by1 - y1) / hh) + fabs((by2 - y2) / hh));
    }
    const double sd = fmin(fmax(1.0 - b3[2] / (double)p.max_depth, 0.1), 1.0);
    const double score = 0.95 * (double)p.scores[i] + 0.05 * (sd + fit) / 2.0;
    const double cls = (double)(p.cls[i] - 1);
    float *o3 = p.box3d + i * 9;
    for (int k = 0; k < 7; ++k) o3[k] = (float)b3[k];
    o3[7] = (float)score;
    o3[8] = (float)cls;
    float *o2 = p.box2d + i * 7;
    o2[0] = (float)by1;
    o2[1] = (float)bx1;
    o2[2] = (float)by2;
    o2[3] = (float)bx2;
    o2[4] = (float)alpha;
    o2[5] = (float)score;
    o2[6] = (float)cls;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------- C ABI

extern "C" {

int mpsr_xyz_map_local_to_global(const float *xyz_local, const float *view_angs, const float *centroids,
                                 float *xyz_global, int b, int p, mpsr_stream_t stream)
{
    MPSR_REQUIRE(b >= 0 && p >= 0, "xyz_map_local_to_global: negative size b=%d p=%d", b, p);
    if (b == 0 || p == 0) return MPSR_OK;
    MPSR_REQUIRE(xyz_local && view_angs && centroids && xyz_global, "xyz_map_local_to_global: null pointer");
    hipLaunchKernelGGL(local_to_global_kernel, dim3(mpsr::ceil_div(p, kThreads), b), dim3(kThreads), 0,
                       mpsr::as_stream(stream), xyz_local, view_angs, centroids, xyz_global, p);
    MPSR_CHECK_LAUNCH("xyz_map_local_to_global");
    return MPSR_OK;
}

int mpsr_xyz_map_local_to_global_grad(const float *grad_global, const float *view_angs, float *grad_local,
                                      float *grad_centroids, int b, int p, mpsr_stream_t stream)
{
    MPSR_REQUIRE(b >= 0 && p >= 0, "xyz_map_local_to_global_grad: negative size b=%d p=%d", b, p);
    if (b == 0) return MPSR_OK;
    MPSR_REQUIRE(grad_global && view_angs, "xyz_map_local_to_global_grad: null pointer");
    hipLaunchKernelGGL(local_to_global_grad_kernel, dim3(b), dim3(kThreads), 0, mpsr::as_stream(stream), grad_global,
                       view_angs, grad_local, grad_centroids, p);
    MPSR_CHECK_LAUNCH("xyz_map_local_to_global_grad");
    return MPSR_OK;
}

int mpsr_proj_err_norm(const float *xyz_global, const float *boxes_2d, const float *cam_p, const float *valid_mask,
                       float *proj_err_maps, float *proj_err_norm, int b, int h, int w, mpsr_stream_t stream)
{
    MPSR_REQUIRE(b >= 0 && h > 0 && w > 0, "proj_err_norm: bad size b=%d h=%d w=%d", b, h, w);
    if (b == 0) return MPSR_OK;
    MPSR_REQUIRE(xyz_global && boxes_2d && cam_p && valid_mask && proj_err_norm, "proj_err_norm: null pointer");
    hipLaunchKernelGGL(proj_err_kernel<0>, dim3(b), dim3(kThreads), 0, mpsr::as_stream(stream), xyz_global, boxes_2d,
                       cam_p, valid_mask, (const float *)nullptr, proj_err_maps, proj_err_norm, h, w);
    MPSR_CHECK_LAUNCH("proj_err_norm");
    return MPSR_OK;
}

int mpsr_proj_err_norm_grad(const float *grad_proj_err_norm, const float *xyz_global, const float *boxes_2d,
                            const float *cam_p, const float *valid_mask, float *grad_xyz_global, int b, int h, int w,
                            mpsr_stream_t stream)
{
    MPSR_REQUIRE(b >= 0 && h > 0 && w > 0, "proj_err_norm_grad: bad size b=%d h=%d w=%d", b, h, w);
    if (b == 0) return MPSR_OK;
    MPSR_REQUIRE(grad_proj_err_norm && xyz_global && boxes_2d && cam_p && valid_mask && grad_xyz_global,
                 "proj_err_norm_grad: null pointer");
    hipLaunchKernelGGL(proj_err_kernel<1>, dim3(b), dim3(kThreads), 0, mpsr::as_stream(stream), xyz_global, boxes_2d,
                       cam_p, valid_mask, grad_proj_err_norm, (float *)nullptr, grad_xyz_global, h, w);
    MPSR_CHECK_LAUNCH("proj_err_norm_grad");
    return MPSR_OK;
}

int mpsr_depth_map_local_to_global(const float *depth_local, int depth_stride, const float *cen_z,
                                   const float *boxes_2d, const float *view_angs, const float *cam_p,
                                   float *depth_global, int b, int h, int w, int rotate_view, mpsr_stream_t stream)
{
    MPSR_REQUIRE(b >= 0 && h > 0 && w > 0 && depth_stride >= 1, "depth_map_local_to_global: bad size b=%d h=%d w=%d",
                 b, h, w);
    if (b == 0) return MPSR_OK;
    MPSR_REQUIRE(depth_local && cen_z && depth_global, "depth_map_local_to_global: null pointer");
    MPSR_REQUIRE(!rotate_view || (boxes_2d && view_angs && cam_p),
                 "depth_map_local_to_global: rotate_view needs boxes_2d, view_angs and cam_p");
    hipLaunchKernelGGL(depth_global_kernel, dim3(mpsr::ceil_div(h * w, kThreads), b), dim3(kThreads), 0,
                       mpsr::as_stream(stream), depth_local, depth_stride, cen_z, boxes_2d, view_angs, cam_p,
                       depth_global, h, w, rotate_view);
    MPSR_CHECK_LAUNCH("depth_map_local_to_global");
    return MPSR_OK;
}

int mpsr_depth_map_local_to_global_grad(const float *grad_global, const float *boxes_2d, const float *view_angs,
                                        const float *cam_p, float *grad_cen_z, int b, int h, int w, int rotate_view,
                                        mpsr_stream_t stream)
{
    MPSR_REQUIRE(b >= 0 && h > 0 && w > 0, "depth_map_local_to_global_grad: bad size b=%d h=%d w=%d", b, h, w);
    if (b == 0) return MPSR_OK;
    MPSR_REQUIRE(grad_global && grad_cen_z, "depth_map_local_to_global_grad: null pointer");
    MPSR_REQUIRE(!rotate_view || (boxes_2d && view_angs && cam_p),
                 "depth_map_local_to_global_grad: rotate_view needs boxes_2d, view_angs and cam_p");
    hipLaunchKernelGGL(depth_global_grad_kernel, dim3(b), dim3(kThreads), 0, mpsr::as_stream(stream), grad_global,
                       boxes_2d, view_angs, cam_p, grad_cen_z, h, w, rotate_view);
    MPSR_CHECK_LAUNCH("depth_map_local_to_global_grad");
    return MPSR_OK;
}

int mpsr_huber_loss_sums(const float *pred, const float *target, const float *weights, int b, int p, int c,
                         float delta, float *sums, float *counts, mpsr_stream_t stream)
{
    MPSR_REQUIRE(b >= 0 && p >= 0 && c >= 1 && delta > 0.f, "huber_loss_sums: bad size b=%d p=%d c=%d", b, p, c);
    if (b == 0) return MPSR_OK;
    MPSR_REQUIRE(pred && target && weights && sums && counts, "huber_loss_sums: null pointer");
    hipLaunchKernelGGL(huber_sum_kernel, dim3(b), dim3(kThreads), 0, mpsr::as_stream(stream), pred, target, weights,
                       p, c, delta, sums, counts);
    MPSR_CHECK_LAUNCH("huber_loss_sums");
    return MPSR_OK;
}

int mpsr_huber_loss_grad(const float *pred, const float *target, const float *weights, const float *scale, int b,
                         int p, int c, float delta, float *grad, mpsr_stream_t stream)
{
    MPSR_REQUIRE(b >= 0 && p >= 0 && c >= 1 && delta > 0.f, "huber_loss_grad: bad size b=%d p=%d c=%d", b, p, c);
    const long long n = (long long)b * p * c;
    if (n == 0) return MPSR_OK;
    MPSR_REQUIRE(pred && target && weights && scale && grad, "huber_loss_grad: null pointer");
    hipLaunchKernelGGL(huber_grad_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       mpsr::as_stream(stream), pred, target, weights, scale, n, c, delta, grad);
    MPSR_CHECK_LAUNCH("huber_loss_grad");
    return MPSR_OK;
}

int mpsr_format_boxes(const float *lwh, const float *view_angs, const float *alpha_bins, const float *alpha_regs,
                      const float *centroids, const float *boxes_2d, const float *scores, const int *class_idx,
                      const float *cam_p, int b, int num_alpha_bins, int img_h, int img_w, int centroid_middle,
                      int post_process_cen_x, float max_depth, float *box_3d, float *box_2d, mpsr_stream_t stream)
{
    MPSR_REQUIRE(b >= 0 && num_alpha_bins >= 1 && img_h > 0 && img_w > 0 && max_depth > 0.f,
                 "format_boxes: bad size b=%d bins=%d image %dx%d", b, num_alpha_bins, img_h, img_w);
    if (b == 0) return MPSR_OK;
    MPSR_REQUIRE(lwh && view_angs && alpha_bins && alpha_regs && centroids && boxes_2d && scores && class_idx &&
                     cam_p && box_3d && box_2d,
                 "format_boxes: null pointer");
    FormatParams fp{lwh,    view_angs, alpha_bins, alpha_regs, centroids,      boxes_2d, scores,          cam_p,
                    class_idx, box_3d, box_2d,     b,          num_alpha_bins, img_h,    img_w,
                    centroid_middle, post_process_cen_x, max_depth};
    hipLaunchKernelGGL(format_boxes_kernel, dim3(mpsr::ceil_div(b, 64)), dim3(64), 0, mpsr::as_stream(stream), fp);
    MPSR_CHECK_LAUNCH("format_boxes");
    return MPSR_OK;
}

}  // extern "C"
