/*
 * monopsr_hip.h -- C ABI of libmonopsr_hip.so: MonoPSR's per-instance hot path on MI355X (gfx950).
 *
 * This is the drop-in boundary.  Every entry point is `extern "C"`, takes plain device pointers and sizes,
 * an explicit HIP stream (passed as void* so this header needs no HIP include), returns an int status
 * (MPSR_OK == 0) and never allocates, frees or synchronises.  Outputs and scratch are caller-allocated; the
 * *_bytes / *_floats helpers say how much.  All tensors are contiguous; floats are fp32, indices int32,
 * images/activations NHWC.
 *
 * Threads: every entry point may be called from any number of threads at once (on different streams, with different
 * scratch); the library keeps no per-call state and the last-error string is thread-local.  The contraction arithmetic
 * and the Winograd policy are OPTIONS OF A CALL since ABI 5 (mpsr_net_opts.math / .winograd_policy, mpsr_conv_opts):
 * they hold for the duration of that entry point on the calling thread only, so concurrent callers with different
 * options do not see each other (the reference's launchers are stateless: tf_nndistance.cpp:168).  mpsr_set_conv_math /
 * mpsr_set_winograd_policy set the process-wide DEFAULTS that calls without options inherit -- atomics read at every
 * launch: change them before starting work, not while another thread is inside an entry point that relies on them.
 *
 * Each declaration cites the reference interface it replaces (paths under /root/reference/src).
 * The first five keep the argument order of the reference's launcher functions so the reference's TF op
 * shells (tf_nndistance.cpp:168,208 / tf_approxmatch.cpp:141-143) could bind to them with the stream appended;
 * INTEGRATION.md shows that binding and the ctypes one used by monopsr_amd.
 */
#ifndef MONOPSR_HIP_H
#define MONOPSR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *mpsr_stream_t; /* a hipStream_t; NULL = the null stream */
typedef void *mpsr_event_t;  /* a hipEvent_t created by the caller */

enum {
    MPSR_OK = 0,
    MPSR_ERR_INVALID_ARG = 1, /* shape / pointer check failed (the reference raises errors::InvalidArgument) */
    MPSR_ERR_HIP = 2,         /* a HIP runtime call or kernel launch failed */
    MPSR_ERR_WORKSPACE = 3,   /* caller-provided scratch is too small */
    MPSR_ERR_UNSUPPORTED = 4  /* configuration outside what the kernels implement */
};

/* Human-readable description of the last non-OK status returned on the calling thread ("" if none). */
const char *mpsr_last_error(void);
/* ABI version, bumped on any signature change. */
int mpsr_abi_version(void);
/* CRC-32C (Castagnoli) of n bytes of HOST memory, continuing from `crc` (0 to start).  The checksum TensorFlow's
 * checkpoint files carry (tensor_bundle block trailers and tensor payloads); used by core/tf_checkpoint.py, which
 * replaces the tf.train.Saver restore of core/checkpoint_utils.py:64-117. */
uint32_t mpsr_crc32c(uint32_t crc, const void *data, size_t n);

/* ------------------------------------------------------------------------------------------------ Chamfer */

/* Replaces NmDistanceKernelLauncher (tf_ops/nn_distance/tf_nndistance.cpp:168, tf_nndistance_g.cu:128-131),
 * i.e. op NnDistance (tf_nndistance.cpp:3-9).  xyz1 (b,n,3), xyz2 (b,m,3) -> dist1 (b,n), idx1 (b,n),
 * dist2 (b,m), idx2 (b,m).  Squared distances; ties resolve to the lowest index; bit-exact with the
 * reference's CPU kernel (tf_nndistance.cpp:21-43).  b, n or m == 0 is a no-op. */
int mpsr_nn_distance_fwd(int b, int n, const float *xyz1, int m, const float *xyz2,
                         float *dist1, int *idx1, float *dist2, int *idx2, mpsr_stream_t stream);

/* Replaces NmDistanceGradKernelLauncher (tf_nndistance.cpp:208, tf_nndistance_g.cu:152-157), op
 * NnDistanceGrad (tf_nndistance.cpp:10-18).  Overwrites grad_xyz1 (b,n,3) and grad_xyz2 (b,m,3) (no
 * pre-zeroing needed).  idx values must lie in [0,m) / [0,n). */
int mpsr_nn_distance_bwd(int b, int n, const float *xyz1, int m, const float *xyz2,
                         const float *grad_dist1, const int *idx1, const float *grad_dist2, const int *idx2,
                         float *grad_xyz1, float *grad_xyz2, mpsr_stream_t stream);

/* ------------------------------------------------------------------------------------------------ EMD */

/* The reference has TWO implementations of the matching, and they differ numerically: its device kernel
 * (tf_approxmatch_g.cu:1-179: 10 annealing levels, fp32 state, match laid out (b,m,n) as tf_approxmatch.py:15-23
 * documents) and its CPU kernel (tf_approxmatch.cpp:23-84, the path BASELINE config 1 runs: 11 levels, double state,
 * the receiver capacity reduced by what was actually taken, match laid out (b,n,m)).  MPSR_EMD_DEVICE is the default
 * and the fast path; MPSR_EMD_HOST reproduces the CPU kernel's arithmetic on the GPU (fp64 state, libm-grade expf) so
 * that outputs can be compared with a TF1-CPU run; a few times slower. */
enum { MPSR_EMD_DEVICE = 0, MPSR_EMD_HOST = 1 };

/* Scratch floats behind `temp` for the full-speed path of the given semantics: per cloud (n+m) * (1 + levels) state
 * words (fp32, or fp64 for MPSR_EMD_HOST) -- the ratios of every level are kept so that match is written exactly once
 * (or never, mpsr_emd_loss).  mpsr_approx_match_temp_floats(b,n,m) == mpsr_emd_temp_floats(b,n,m,MPSR_EMD_DEVICE). */
size_t mpsr_emd_temp_floats(int b, int n, int m, int semantics);
size_t mpsr_approx_match_temp_floats(int b, int n, int m);

/* Replaces approxmatchLauncher (tf_approxmatch.cpp:141, tf_approxmatch_g.cu:1-182), op ApproxMatch
 * (tf_approxmatch.cpp:7-10): the launcher's arguments, then the size of `temp` in floats and the stream.
 * xyz1 (b,n,3), xyz2 (b,m,3) -> match (b,m,n), device-kernel semantics.
 *   temp_floats >= mpsr_approx_match_temp_floats(b,n,m): fast path (match written once);
 *   temp_floats >= b*(n+m)*2, what the reference's op shell allocates (tf_approxmatch.cpp:168): same result bit for
 *     bit.  The per-level state then lives in the tail of each cloud's own block of `match` until those rows are
 *     written last (same passes, same single write of match: as fast as the fast path); clouds too small or too
 *     ragged for that (n*m < 11*(n+m)+n, or more than 64 rows of state) accumulate match level by level as the
 *     reference does.  `match` must not be read by anyone else while the call runs (it never could be);
 *   less: MPSR_ERR_WORKSPACE, nothing is written. */
int mpsr_approx_match(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                      size_t temp_floats, mpsr_stream_t stream);

/* The same with the semantics chosen.  MPSR_EMD_HOST: match is (b,n,m) like the CPU kernel's output, temp must hold
 * mpsr_emd_temp_floats(b,n,m,MPSR_EMD_HOST) floats and be 8-byte aligned.  (mpsr_match_cost / mpsr_match_cost_grad
 * take a (b,n,m) match when called with the clouds swapped: cost(b, m, n, xyz2, xyz1, match), and
 * grad(b, m, n, xyz2, xyz1, match, grad2, grad1).) */
int mpsr_approx_match_ex(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                         size_t temp_floats, int semantics, mpsr_stream_t stream);

/* EarthMoversDistance (losses_custom.py:135-165: approx_match -> match_cost, gradient by MatchCostGrad with match
 * held constant) in one call that never materialises match: cost (b), grad1 = d cost / d xyz1 (b,n,3),
 * grad2 (b,m,3).  Every match entry is recomputed from the per-level ratios where it is consumed; results equal
 * mpsr_approx_match + mpsr_match_cost + mpsr_match_cost_grad to fp32 summation order.  grad1 / grad2 may be NULL
 * (cost only: the metric of monopsr_model.py:1143-1149).  temp: mpsr_emd_temp_floats(b,n,m,semantics) floats, or --
 * ABI 6 -- mpsr_emd_loss_temp_floats(b,n,m,semantics): with that much scratch (the state + both clouds re-ordered +
 * their permutations: b*(n+m)*4 floats more; device semantics, clouds of up to 4096 points) the call CAN cull levels:
 * sort both clouds into Morton order and leave out, chunk of 32 points by chunk, the pairs whose exponential
 * exp(level * d^2) is exactly zero in fp32 at the four steepest levels (-16384 .. -256: d beyond 0.08 .. 0.64) -- the
 * reference's kernels evaluate every pair at every level (tf_approxmatch_g.cu:21-160).  What is skipped is exactly zero
 * (bit-identical to the unculled evaluation of the sorted clouds), but the sums run in the order of the SORTED clouds,
 * and the annealing's clamps amplify that rounding difference: isolated gradient elements move by up to ~1e-3 of the
 * largest against the evaluation in the caller's order.  The form is therefore OFF unless switched on
 * (mpsr_debug_set_emd_cull(1): -4 .. -8 % of the call at 256 x 2048^2, slower below ~64 clouds); by default the call
 * ignores the extra scratch.  Gradients come back in the caller's point order either way. */
size_t mpsr_emd_loss_temp_floats(int b, int n, int m, int semantics);
int mpsr_emd_loss(int b, int n, int m, const float *xyz1, const float *xyz2, float *cost, float *grad1, float *grad2,
                  float *temp, size_t temp_floats, int semantics, mpsr_stream_t stream);

/* Replaces matchcostLauncher (tf_approxmatch.cpp:142, tf_approxmatch_g.cu:183-228), op MatchCost.
 * -> out (b). */
int mpsr_match_cost(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *out,
                    mpsr_stream_t stream);

/* Replaces matchcostgradLauncher (tf_approxmatch.cpp:143, tf_approxmatch_g.cu:229-295), op MatchCostGrad.
 * -> grad1 (b,n,3), grad2 (b,m,3); both fully overwritten. */
int mpsr_match_cost_grad(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                         float *grad1, float *grad2, mpsr_stream_t stream);

/* ------------------------------------------------------------------------------------------------ image ops */

/* tf.image.crop_and_resize (bilinear, extrapolation_value) as called at monopsr_model.py:222-226 and
 * net_builder.py:54-59.  image (nimg,H,W,C), boxes (nb,4) = [y1,x1,y2,x2] normalised, box_ind (nb) or NULL
 * (= all 0) -> out (nb,ch,cw,C). */
int mpsr_crop_and_resize(const float *image, int nimg, int H, int W, int C, const float *boxes,
                         const int *box_ind, int nb, int ch, int cw, float extrapolation_value, float *out,
                         mpsr_stream_t stream);

/* tf.image.resize_bilinear (TF-1.8 legacy kernel, no half-pixel centres) as called at monopsr_model.py:228-233,
 * net_builder.py:72-83 and img_preprocessor.py:32.  in (B,H,W,C) -> out (B,OH,OW,C). */
int mpsr_resize_bilinear(const float *in, int B, int H, int W, int C, int OH, int OW, int align_corners,
                         float *out, mpsr_stream_t stream);

/* slim.max_pool2d: window k, stride s, padding SAME (pad_same=1, TF pads bottom/right first) or VALID.
 * resnet_v1.py:235 (3x3/2 SAME), net_builder.py:60,68 (2x2/2 VALID). */
int mpsr_max_pool(const float *in, int B, int H, int W, int C, int k, int s, int pad_same, float *out,
                  mpsr_stream_t stream);

/* ------------------------------------------------------------------------------------------------ conv / FC */

/* One stride-1 SAME convolution (optionally atrous) or fully-connected layer as an fp32-MFMA implicit GEMM with
 * a fused epilogue: y = act(conv(x, w) + bias + residual).  Serves every slim.conv2d / slim.fully_connected on
 * the path (resnet_v1.py:116-127, resnet_utils.py:112, net_builder.py:67,77,85, monopsr_output_builder.py:101,
 * 140,166,253,...) with BatchNorm folded into w/bias by the caller.
 *   x        (B,H,W,C) NHWC, C % 4 == 0, 16-byte aligned
 *   w        (N, KH*KW*C) row-major: w[n][(ky*KW+kx)*C + c]  (the TF HWIO tensor transposed to O,HWI)
 *   bias     (N) or NULL;  residual (B,H,W,N) or NULL;  relu 0/1
 *   y        (B,H,W,N)
 *   split_k  >= 1: one output tile per workgroup; when > 1 the K loop is cut into split_k slices, `ws` must hold
 *            split_k*B*H*W*N floats (partial sums, reduced by a second kernel).
 *            0: the library schedules the launch itself -- stream-K (persistent workgroups that each take the same
 *            number of K steps of the launch's tile sequence; partial tiles meet in `ws`) when `ws` holds
 *            mpsr_conv2d_scratch_floats(B,H,W,N) floats, otherwise as split_k = 1.  Results are deterministic for
 *            a given shape and device, and equal to split_k = 1 up to fp32 summation order on the split tiles.
 *            With split_k = 0 and that scratch, large dense 3x3 layers (dilation 1, no residual, even H and W,
 *            C % 16 == 0, C and N >= 64, B*H*W >= 65536: the map decoder) run as Winograd F(2x2,3x3) -- 16
 *            products per 2x2 output tile where the direct form has 36; fp32 results agree with the direct form to
 *            ~1e-6 relative (tests/test_net_gpu.py checks 1e-5 against fp64).
 * A fully-connected layer is H=W=KH=KW=1.
 * mpsr_conv2d_plan reports, for a layer left to the library (split_k = 0, scratch given), which kernel serves it
 * (*kind: 0 implicit GEMM, 1 Winograd F(2x2,3x3), 2 direct narrow-N kernel, 3 Winograd F(4x4,3x3), 4 Winograd
 * F(3x3,3x3) -- or its sixteen-product form where a sub-grid is one zero-padded tile -- on the sub-grids of an atrous layer, 5 persistent pointwise kernel, 6 few-row fully-connected kernel) and the multiply-add FLOPs that kernel issues --
 * for throughput accounting (bench.py's roofline.executed), not needed to run anything. */
size_t mpsr_conv2d_scratch_floats(int B, int H, int W, int N);
int mpsr_conv2d_plan(int B, int H, int W, int C, int N, int KH, int KW, int dilation, int *kind,
                     double *executed_flops);
int mpsr_conv2d_nhwc_f32(const float *x, int B, int H, int W, int C, const float *w, const float *bias,
                         const float *residual, float *y, int N, int KH, int KW, int dilation, int relu,
                         int split_k, float *ws, size_t ws_floats, mpsr_stream_t stream);

/* Options of ONE call (ABI 5).  The reference's launchers carry no state (tf_nndistance.cpp:168); here the
 * arithmetic mode and the Winograd policy have process-wide defaults (mpsr_set_conv_math / mpsr_set_winograd_policy
 * below) and every entry point that takes options can override them for the duration of the call -- calls on different
 * host threads with different options do not see each other (tests/test_threads_gpu.py).  A zero-initialised struct
 * inherits the defaults. */
enum { MPSR_CALL_MATH_INHERIT = 0, MPSR_CALL_MATH_FP32 = 1, MPSR_CALL_MATH_BF16X3 = 2 };
enum { MPSR_CALL_WINOGRAD_INHERIT = 0, MPSR_CALL_WINOGRAD_AUTO = 1, MPSR_CALL_WINOGRAD_OFF = 2,
       MPSR_CALL_WINOGRAD_ACCURATE = 3 /* ABI 6 */ };
typedef struct mpsr_conv_opts {
    int32_t math;            /* MPSR_CALL_MATH_* */
    int32_t winograd_policy; /* MPSR_CALL_WINOGRAD_* */
} mpsr_conv_opts;
int mpsr_conv2d_nhwc_f32_ex(const float *x, int B, int H, int W, int C, const float *w, const float *bias,
                            const float *residual, float *y, int N, int KH, int KW, int dilation, int relu,
                            int split_k, float *ws, size_t ws_floats, const mpsr_conv_opts *opts, mpsr_stream_t stream);

/* Arithmetic of every contraction mpsr_conv2d_nhwc_f32 and the network entry points run (process-wide DEFAULT;
 * per call: mpsr_conv_opts / mpsr_net_opts).
 *   MPSR_MATH_FP32   (default) exact fp32 products on v_mfma_f32_32x32x2_f32: what every parity and benchmark
 *                    number of this library refers to unless it says otherwise.
 *   MPSR_MATH_BF16X3 opt-in fast mode: operands split into hi + lo bfloat16 halves on the fly, each product evaluated
 *                    as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (~2^-16 relative
 *                    error per product; inputs, outputs and weights stay fp32 in memory).
 * The reference has no such switch (TF-1.8 convolutions are fp32); the path's drift budget is 1e-3 (BASELINE.json). */
enum { MPSR_MATH_FP32 = 0, MPSR_MATH_BF16X3 = 1 };
int mpsr_set_conv_math(int mode);
int mpsr_get_conv_math(void);

/* Whether 3x3 layers may run in a Winograd transform domain (process-wide DEFAULT, like the arithmetic above; per call:
 * mpsr_conv_opts / mpsr_net_opts).
 *   MPSR_WINOGRAD_AUTO (default) block3's atrous layers (one zero-padded 3x3 tile per pixel sub-grid) in sixteen
 *                      products per tile, F(3x3,3x3) on the tiles-with-halos of blocks 1-2, F(4x4,3x3) / F(2x2,3x3) on
 *                      the dense decoder layers wherever they are faster.  Error against float64 of the tensor's
 *                      SCALE: 3e-7 / 3e-6 / 1.5e-5 / 1e-6 (a direct fp32 convolution: 2.5e-7 .. 5e-7) -- far inside the
 *                      path's 1e-3 budget.  The transforms mix the values of a whole input patch, so a SMALL output
 *                      next to a very large activation carries an error relative to the large one: on maps with 1 % of
 *                      the entries 1000x the rest, elements down to 1e-3 of the tensor's maximum were measured
 *                      2e-3 .. 4e-3 off relative to THEMSELVES under F(4x4,3x3), 0.5e-3 .. 2e-3 under F(3x3,3x3) with
 *                      halos, 2e-4 in the sixteen-product form (the direct kernels 1e-4;
 *                      tests/test_hostile_inputs_gpu.py).  The decoder end to end on such features (100x outliers):
 *                      1.2e-3, where fp32 arithmetic without any Winograd kernel measures 8.6e-4.
 *   MPSR_WINOGRAD_OFF  direct / implicit-GEMM kernels everywhere (the upsampled convolutions keep their exact tap
 *                      GEMM): element-wise 5e-5 .. 1e-4 on the same inputs, at 1.33x the step time (17.8 vs 13.4 ms:
 *                      bench.py's `winograd_off_mode` object times it and measures both policies' errors).  For callers whose
 *                      activations are heavy-tailed AND who read small outputs individually.
 *   MPSR_WINOGRAD_ACCURATE (r06) only the transform-domain forms whose ELEMENT-WISE error on such maps stays inside
 *                      1e-3: the sixteen-product form for block3's atrous layers (2e-4) and F(2x2,3x3) for the dense
 *                      decoder layers (transform constants 1 and 1/2); F(4x4,3x3) and the F(3x3,3x3) halo tiles are not
 *                      used (their layers take F(2x2,3x3) / the direct kernels).  Between the two above in speed
 *                      (bench.py's `winograd_accurate_mode`). */
enum { MPSR_WINOGRAD_AUTO = 0, MPSR_WINOGRAD_OFF = 1, MPSR_WINOGRAD_ACCURATE = 2 };
int mpsr_set_winograd_policy(int policy);
int mpsr_get_winograd_policy(void);

/* tf.image.resize_bilinear (align_corners as given) followed by a 3x3 SAME slim.conv2d, as the map decoder applies
 * them (monopsr/builders/net_builder.py:72-77, :81-85), WITHOUT forming the upsampled map: a 1x1 GEMM of the source
 * map with 9 N outputs (channel mixing commutes with the per-channel upsampling) + a 9-tap x 4-corner gather
 * (csrc/upconv.hip).  x (B,h,w,C) -> y (B,OH,OW,N) NHWC; weights (N, 9 C) as for mpsr_conv2d_nhwc_f32.  Needs
 * N % 128 == 0, C % 64 == 0, C >= 128 (MPSR_ERR_UNSUPPORTED otherwise: use mpsr_resize_bilinear + mpsr_conv2d_nhwc_f32)
 * and mpsr_conv3x3_upsampled_scratch_floats() floats of scratch.  fp32 arithmetic whatever mpsr_set_conv_math says. */
size_t mpsr_conv3x3_upsampled_scratch_floats(int B, int h, int w, int C, int N);
int mpsr_conv3x3_upsampled_f32(const float *x, int B, int h, int w, int C, int OH, int OW, int align_corners,
                               const float *weights, const float *bias, int relu, float *y, int N, float *ws,
                               size_t ws_floats, mpsr_stream_t stream);
/* 0: the shape is outside what mpsr_conv3x3_upsampled_f32 takes; 1: forward only; 2: forward and backward. */
int mpsr_conv3x3_upsampled_applies(int B, int h, int w, int C, int OH, int OW, int N, int align_corners);
/* Its backward for training (TF autodiff of the two operators in the reference): dy (B,OH,OW,N) = gradient at the
 * convolution's output BEFORE bias / activation.  dw (N, 9 C) += weight gradient (accumulated, like
 * mpsr_conv2d_wgrad_f32; the bias gradient is mpsr_bias_grad of dy); dx (B,h,w,C) = data gradient, or NULL.  Everything
 * happens at the SOURCE resolution: dz = the transposed gather of dy, then a 1x1 weight gradient and a 1x1 GEMM with
 * K = 9 N -- in place of the resize gradient + the 3x3 data and weight gradients on the upsampled map.  Deterministic
 * except for the weight gradient's pixel slices (fp32 atomics, as mpsr_conv2d_wgrad_f32). */
size_t mpsr_conv3x3_upsampled_bwd_scratch_floats(int B, int h, int w, int C, int N);
int mpsr_conv3x3_upsampled_bwd_f32(const float *x, const float *dy, int B, int h, int w, int C, int OH, int OW,
                                   int align_corners, const float *weights, int N, float *dw, float *dx, float *ws,
                                   size_t ws_floats, mpsr_stream_t stream);

/* Explicit im2col for the ResNet root: explicit zero pad 3 + 7x7 stride-2 VALID (resnet_utils.py:115-122 via
 * resnet_v1.py:234).  x (B,H,W,3) -> cols (B*OH*OW, kpad) with OH=(H+6-7)/2+1; column (ky*7+kx)*3+c, columns
 * 147..kpad-1 zero.  kpad % 32 == 0, kpad >= 147. */
int mpsr_im2col_root(const float *x, int B, int H, int W, float *cols, int kpad, mpsr_stream_t stream);

/* ------------------------------------------------------------------------------------------------ backward */
/* The reference gets gradients from TensorFlow autodiff (core/trainer.py:71-81, builders/optimizer_builder.py:61-80);
 * these entry points are what a training step of the instance path needs beyond the forward kernels. */

/* dW[n][(ky*KW+kx)*C + c] += sum over pixels of dy[pixel][n] * x[pixel + tap][c] (same geometry as
 * mpsr_conv2d_nhwc_f32).  x (B,H,W,C), dy (B,H,W,N), dw (N, KH*KW*C) must be zeroed by the caller (partial sums
 * from pixel slices are combined with fp32 atomics).  C % 4 == 0, N % 4 == 0.
 * db (N) or NULL: the bias gradient db[n] += sum over pixels of dy[pixel][n] rides along (column sums of the dy
 * tiles the kernel stages anyway; also pre-zeroed by the caller). */
int mpsr_conv2d_wgrad_f32(const float *x, const float *dy, int B, int H, int W, int C, int N, int KH, int KW,
                          int dilation, float *dw, float *db, mpsr_stream_t stream);
/* The same with caller scratch: `ws` of at least mpsr_conv2d_wgrad_scratch_floats(...) floats (0 when the layer does
 * not use any) lets the big dense 3x3 layers (the map decoder: stride 1, no dilation, H and W multiples of 4, C and N
 * multiples of 32, B*H*W >= 65536) run in the Winograd F(4x4,3x3) domain -- 36 products per channel pair and 4x4
 * block where the direct weight gradient has 144; results agree with mpsr_conv2d_wgrad_f32 to ~1e-4 of the
 * gradient's scale.  ws = NULL or too small: exactly mpsr_conv2d_wgrad_f32.
 * Run-to-run reproducibility: BOTH forms combine the partial sums of their pixel / tile slices with fp32 atomics, so a
 * weight gradient's last bits depend on the order in which workgroups arrive (relative spread ~1e-7 direct, ~1e-6 in
 * the Winograd domain, where the transformed sums are larger); forward results and data gradients are deterministic. */
size_t mpsr_conv2d_wgrad_scratch_floats(int B, int H, int W, int C, int N, int KH, int KW, int dilation);
int mpsr_conv2d_wgrad_ws_f32(const float *x, const float *dy, int B, int H, int W, int C, int N, int KH, int KW,
                             int dilation, float *dw, float *db, float *ws, size_t ws_floats, mpsr_stream_t stream);

/* Weight re-layout for the data gradient: wd[c][((KH*KW-1-t)*N) + n] = w[n][t*C + c].  Then
 * dx = mpsr_conv2d_nhwc_f32(dy, ..., w = wd, N := C, C := N) with the same KH, KW, dilation. */
int mpsr_conv2d_dgrad_pack(const float *w, int N, int KH, int KW, int C, float *wd, mpsr_stream_t stream);

/* The re-layout of ALL the layers of a training step in one launch (the reference's step differentiates every layer of
 * the graph in one session.run: core/trainer.py:76-81, builders/optimizer_builder.py:61-80; here the weights only
 * change in the optimizer step, so their data-gradient layouts are made once per step, not once per layer launch).
 * A job = one layer: w (N, T = KH*KW, C) -> wd (C, T, Nd) with Nd >= N (a filter padded to Nd output channels: rows
 * N..Nd-1 read as zero).  The caller describes the jobs once (`chunk0` is filled in by the library), builds the table
 * on the host, copies its mpsr_dgrad_pack_table_bytes() bytes to device memory and from then on issues
 * mpsr_conv2d_dgrad_pack_batch(table_dev, n_chunks) whenever the weights changed.  Pointers inside the table are
 * device pointers and must stay valid; the table holds no other state. */
typedef struct mpsr_pack_job {
    const float *w;
    float *wd;
    int32_t N, Nd, T, C;
    int64_t chunk0;
} mpsr_pack_job;
size_t mpsr_dgrad_pack_table_bytes(const mpsr_pack_job *jobs, int n_jobs); /* 0 = bad jobs */
int mpsr_dgrad_pack_table_build(const mpsr_pack_job *jobs, int n_jobs, void *table_host, long long *n_chunks);
int mpsr_conv2d_dgrad_pack_batch(const void *table_dev, long long n_chunks, mpsr_stream_t stream);

/* Fused activation + bias gradient in one pass over dy (M,N): g = (y > 0 ? dy : 0) when y != NULL else dy;
 * dx = g when dx != NULL (may alias dy); db[n] += sum_m g[m][n] when db != NULL (zeroed by the caller). */
int mpsr_act_bias_grad(const float *dy, const float *y, float *dx, float *db, long long M, int N,
                       mpsr_stream_t stream);

/* db[n] += sum_m dy[m][n]; db zeroed by the caller. */
int mpsr_bias_grad(const float *dy, long long M, int N, float *db, mpsr_stream_t stream);

/* dx = dy where y > 0 else 0 (y = post-activation output of the layer).  dx may alias dy. */
int mpsr_relu_grad(const float *dy, const float *y, float *dx, long long total, mpsr_stream_t stream);

/* The ReLU mask of a tensor as bits, and a 1x1 data gradient that leaves through it (training: a bottleneck unit's
 * input gradient, TF autodiff of resnet_v1.py:104-135, without the elementwise ReLU-gradient pass over it).
 * Layout: bits[(m >> 5) * N + n], bit b <-> row m = 32 (m >> 5) + (b & 3) + 8 ((b >> 2) & 3) + 4 (b >> 4) -- the
 * accumulator layout of the 32x32 MFMA, so that a convolution can write the words from its epilogue;
 * mpsr_relu_bitmask_words(M, N) = ceil(M / 32) * N words, 16-byte aligned; N % 4 == 0.
 * mpsr_relu_bitmask: the words of y (M,N): bit = (y[m][n] > 0), rows >= M zero.
 * mpsr_conv1x1_relu_bitmask_f32: y = act(x w^T + bias + residual) on the persistent pointwise kernel AND the words
 * of y in the same launch (bits of rows >= M unspecified); y is bit-identical to mpsr_conv2d_nhwc_f32's.
 * mpsr_conv1x1_masked_f32: y[m][n] = bit(m, n) ? sum_k x[m][k] w[n][k] + bias[n] + residual[m][n] : 0 (bias,
 * residual may be NULL); kept elements are bit-identical to the unmasked launch.
 * mpsr_conv1x1_masked_applies says whether the shape (and the arithmetic mode: fp32 only) is taken by the last two --
 * otherwise MPSR_ERR_UNSUPPORTED and the caller runs mpsr_conv2d_nhwc_f32 (+ mpsr_relu_grad). */
/* y = conv(x, w) (no bias, as scheduled by the library: split_k = 0 of mpsr_conv2d_nhwc_f32, same scratch) where
 * mask > 0 and zero elsewhere; mask (B,H,W,N) = the post-ReLU tensor this data gradient belongs to.  The F(3x3,3x3)
 * atrous kernel (block3's conv2) applies it in its epilogue, other shapes run mpsr_relu_grad in place afterwards. */
int mpsr_conv2d_relu_masked_f32(const float *x, int B, int H, int W, int C, const float *w, const float *mask, float *y,
                                int N, int KH, int KW, int dilation, float *ws, size_t ws_floats, mpsr_stream_t stream);
long long mpsr_relu_bitmask_words(long long M, int N);
int mpsr_relu_bitmask(const float *y, long long M, int N, unsigned *bits, mpsr_stream_t stream);
int mpsr_conv1x1_masked_applies(long long M, int K, int N);
int mpsr_conv1x1_relu_bitmask_f32(const float *x, long long M, int K, const float *w, const float *bias,
                                  const float *residual, int relu, float *y, unsigned *bits, int N, mpsr_stream_t stream);
int mpsr_conv1x1_masked_f32(const float *x, long long M, int K, const float *w, const float *bias,
                            const float *residual, const unsigned *mask, float *y, int N, mpsr_stream_t stream);

/* Gradient of mpsr_max_pool w.r.t. its input (first maximum of each window takes the gradient); dx (B,H,W,C) is
 * fully overwritten. */
int mpsr_max_pool_grad(const float *x, const float *dy, int B, int H, int W, int C, int k, int s, int pad_same,
                       float *dx, mpsr_stream_t stream);

/* Gradient of mpsr_resize_bilinear w.r.t. its input; dy (B,OH,OW,C) -> dx (B,H,W,C), fully overwritten. */
int mpsr_resize_bilinear_grad(const float *dy, int B, int H, int W, int C, int OH, int OW, int align_corners,
                              float *dx, mpsr_stream_t stream);

/* One Adam step over flat fp32 buffers (tf.train.AdamOptimizer update rule, optimizer_builder.py:61-80):
 * g' = grad*grad_scale; m = b1*m+(1-b1)*g'; v = b2*v+(1-b2)*g'^2; p -= lr*sqrt(1-b2^step)/(1-b1^step) * m/(sqrt(v)+eps). */
int mpsr_adam_step(float *param, const float *grad, float *m, float *v, long long n, float lr, float beta1,
                   float beta2, float eps, int step, float grad_scale, mpsr_stream_t stream);
/* The same update with the bias-corrected rate lr_t = lr*sqrt(1-b2^step)/(1-b1^step) read from DEVICE memory when the
 * kernel runs: a launch captured into a HIP graph (the training step replayed as one graph launch) then follows the
 * schedule -- the caller writes *lr_t_dev (stream-ordered) before each replay. */
int mpsr_adam_step_lr_dev(float *param, const float *grad, float *m, float *v, long long n, const float *lr_t_dev,
                          float beta1, float beta2, float eps, float grad_scale, mpsr_stream_t stream);

/* Gradient of mpsr_crop_and_resize w.r.t. the image (TensorFlow's CropAndResizeGradImage, reached through autodiff
 * from net_builder.py:54-59 when the full-image trunk trains).  grad_out (nb,ch,cw,C) -> grad_image (nimg,H,W,C),
 * zeroed inside and accumulated with fp32 atomics; samples that were extrapolated contribute nothing. */
int mpsr_crop_and_resize_grad(const float *grad_out, int nimg, int H, int W, int C, const float *boxes,
                              const int *box_ind, int nb, int ch, int cw, float *grad_image, mpsr_stream_t stream);

/* Training-mode batch normalisation of the map decoder (net_builder.py:76-87: slim.batch_norm with is_training,
 * statistics over (N,H,W) per channel, no scale, epsilon 1e-3, then ReLU).  z is the (M, C) convolution output,
 * M = N*H*W, C % 4 == 0, C <= 1024.  Statistics are accumulated in fp64 around the first row for conditioning:
 *   sum[c] = sum_m (z[m][c] - z[0][c]),  sumsq_shifted[c] = sum_m (z[m][c] - z[0][c])^2
 * so mean = z[0] + sum/M and the (biased) variance = sumsq_shifted/M - (sum/M)^2. */
int mpsr_batch_norm_stats(const float *z, long long M, int C, double *sum, double *sumsq_shifted,
                          mpsr_stream_t stream);
/* ABI 6: the per-channel arithmetic between the passes as one launch each (they were ~16 and ~4 tiny tensor operations
 * per layer and step on the host framework's side).  Forward: d = sum / M, mean = z_row0 + d (z_row0 = the first row of
 * z, the shift of the sums), var = max(sumsq_shifted / M - d^2, 0) in fp64; mean and inv_std = 1 / sqrt(var + eps) as
 * floats; moving_mean = moving_mean * decay + (1 - decay) * mean and moving_variance likewise with the UNBIASED
 * variance var * M / max(M - 1, 1), as TensorFlow's fused kernel feeds its moving average (either may be NULL: not
 * updated).  Backward: dbeta += sum_g (NULL: not deposited), mean_g = sum_g / count, mean_gz = sum_gz / count (count =
 * the rows the statistics were taken over).  A caller that pools the sums over ranks does this arithmetic itself. */
int mpsr_batch_norm_finalize(const double *sum, const double *sumsq_shifted, const float *z_row0, long long M, int C,
                             float eps, float decay, float *moving_mean, float *moving_variance, float *mean,
                             float *inv_std, mpsr_stream_t stream);
int mpsr_batch_norm_grad_finalize(const double *sum_g, const double *sum_gz, double count, int C, float *dbeta,
                                  float *mean_g, float *mean_gz, mpsr_stream_t stream);
/* y = act((z - mean) * inv_std + beta); relu 0/1; mean, inv_std, beta (C). */
int mpsr_batch_norm_apply(const float *z, long long M, int C, const float *mean, const float *inv_std,
                          const float *beta, int relu, float *y, mpsr_stream_t stream);
/* Backward, pass 1: with g = dy * (y > 0) (g = dy when y == NULL) and zhat = (z - mean) * inv_std:
 * sum_g[c] = sum_m g (= the beta gradient), sum_gz[c] = sum_m g * zhat; fp64, zeroed inside. */
int mpsr_batch_norm_grad_sums(const float *dy, const float *y, const float *z, long long M, int C, const float *mean,
                              const float *inv_std, double *sum_g, double *sum_gz, mpsr_stream_t stream);
/* Backward, pass 2: dz = inv_std * (g - mean_g - zhat * mean_gz), mean_g = sum_g / M, mean_gz = sum_gz / M. */
int mpsr_batch_norm_grad(const float *dy, const float *y, const float *z, long long M, int C, const float *mean,
                         const float *inv_std, const float *mean_g, const float *mean_gz, float *dz,
                         mpsr_stream_t stream);

/* ABI 6: the same two passes for a layer with ReLU WITHOUT reading y: the mask y > 0 is rebuilt from z as
 * relu((z - mean) * inv_std + beta) > 0 with mpsr_batch_norm_apply's own fused multiply-add (the same bits), which takes
 * one of the three (four) tensor-sized streams out of each pass. */
int mpsr_batch_norm_grad_sums_z(const float *dy, const float *z, long long M, int C, const float *mean,
                                const float *inv_std, const float *beta, double *sum_g, double *sum_gz,
                                mpsr_stream_t stream);
int mpsr_batch_norm_grad_z(const float *dy, const float *z, long long M, int C, const float *mean, const float *inv_std,
                           const float *beta, const float *mean_g, const float *mean_gz, float *dz, mpsr_stream_t stream);

/* tf.clip_by_norm applied to every variable of a flat gradient buffer separately, as
 * slim.learning.create_train_op(clip_gradient_norm=1.0) does (core/trainer.py:78-81): g *= clip / max(||g||, clip).
 * The caller describes the variables once as a chunk table (device arrays): chunk i covers
 * grads[chunk_begin[i] .. +chunk_len[i]) and belongs to variable chunk_seg[i] in [0, n_segments).
 * chunk_seg ascends (a variable's chunks are consecutive).
 * sumsq: scratch of sumsq_floats >= n_segments + n_chunks floats (MPSR_ERR_WORKSPACE otherwise): the first n_segments
 * hold each variable's squared norm on return, the rest the chunks' partial sums.  ABI 6 (signature changed: sumsq_floats):
 * the norms are summed in a FIXED order -- per chunk, then over a variable's chunks -- with no atomics, so every replica of
 * a data-parallel run scales the same reduced gradient by the same bits and the replicas' parameters stay bit-identical
 * (tests/test_sharded_training_step_gpu.py; an atomic accumulation left two ranks one ulp apart after one step). */
int mpsr_clip_by_norm_segments(float *grads, const int *chunk_seg, const long long *chunk_begin,
                               const int *chunk_len, int n_chunks, float *sumsq, size_t sumsq_floats, int n_segments,
                               float clip_norm, mpsr_stream_t stream);

/* ABI 6: the tail of a training step in three launches -- per-variable tf.clip_by_norm (core/trainer.py:78-81), the Adam
 * update (builders/optimizer_builder.py:61-80) and the parameter moving average (tf.contrib.opt.MovingAverageOptimizer,
 * optimizer_builder.py:75-80) over the same chunk table: squared norms, then ONE pass that scales the gradient on its way
 * into the update (grads are left as they came), updates m / v / param and, when `shadow` is not NULL,
 * shadow += (1 - ema_decay) * (param - shadow).  Operation by operation the arithmetic of mpsr_clip_by_norm_segments ->
 * mpsr_adam_step -> that moving average; elements outside every chunk (alignment padding) are not touched.
 * clip_norm <= 0: no clipping (sumsq may be NULL).  sumsq / sumsq_floats as mpsr_clip_by_norm_segments (deterministic
 * norms).  step = 1-based Adam step (bias correction). */
int mpsr_clip_adam_ema_step(float *param, const float *grad, float *m, float *v, float *shadow, const int *chunk_seg,
                            const long long *chunk_begin, const int *chunk_len, int n_chunks, float *sumsq,
                            size_t sumsq_floats, int n_segments, float clip_norm, float lr, float beta1, float beta2,
                            float eps, int step, float ema_decay, mpsr_stream_t stream);

/* ------------------------------------------------------------------------- per-box geometry and map losses
 * SURVEY.md 8(f) rows 3-4.  Maps are (b, h, w, c) row-major; p = h*w points per instance. */

/* instance_utils.py:567-602 tf_inst_xyz_map_local_to_global: xyz_global = T(centroid) R_y(view_ang) xyz_local.
 * xyz_local/xyz_global (b,p,3); view_angs (b); centroids (b,3). */
int mpsr_xyz_map_local_to_global(const float *xyz_local, const float *view_angs, const float *centroids,
                                 float *xyz_global, int b, int p, mpsr_stream_t stream);

/* Its gradient: grad_local (b,p,3) = R^T grad_global, grad_centroids (b,3) = sum over points; either may be NULL. */
int mpsr_xyz_map_local_to_global_grad(const float *grad_global, const float *view_angs, float *grad_local,
                                      float *grad_centroids, int b, int p, mpsr_stream_t stream);

/* monopsr_output_builder.py:681-746 get_proj_err_maps_norm: (expected pixel-centre grid of the 2-D box - projection
 * of xyz_global by cam_p) / box [w,h], * valid_mask, clipped to [-2,2] -> proj_err_maps (b,h,w,2) (may be NULL);
 * proj_err_norm (b) = sum over the map / max(sum(valid_mask), 1).  boxes_2d (b,4) [y1,x1,y2,x2]; cam_p (12);
 * valid_mask (b,h,w). */
int mpsr_proj_err_norm(const float *xyz_global, const float *boxes_2d, const float *cam_p, const float *valid_mask,
                       float *proj_err_maps, float *proj_err_norm, int b, int h, int w, mpsr_stream_t stream);

/* d(sum_b grad_proj_err_norm[b] * proj_err_norm[b]) / d xyz_global -> grad_xyz_global (b,h,w,3), overwritten. */
int mpsr_proj_err_norm_grad(const float *grad_proj_err_norm, const float *xyz_global, const float *boxes_2d,
                            const float *cam_p, const float *valid_mask, float *grad_xyz_global, int b, int h, int w,
                            mpsr_stream_t stream);

/* instance_utils.py:605-680 tf_inst_depth_map_local_to_global: depth_global (b,h,w) = depth_local + cen_z[b]
 * (+ the view-normalisation offset, linear along the ROW axis as the reference lays it out, when rotate_view).
 * depth_local is read with a stride of depth_stride floats (3 to read the z channel of an xyz map in place). */
int mpsr_depth_map_local_to_global(const float *depth_local, int depth_stride, const float *cen_z,
                                   const float *boxes_2d, const float *view_angs, const float *cam_p,
                                   float *depth_global, int b, int h, int w, int rotate_view, mpsr_stream_t stream);

/* Gradient w.r.t. cen_z (b); the gradient w.r.t. depth_local is grad_global itself. */
int mpsr_depth_map_local_to_global_grad(const float *grad_global, const float *boxes_2d, const float *view_angs,
                                        const float *cam_p, float *grad_cen_z, int b, int h, int w, int rotate_view,
                                        mpsr_stream_t stream);

/* Per-instance sums behind WeightedNonZeroSmoothL1LocalizationLoss (losses_custom.py:93-132, tf.losses.huber_loss
 * with Reduction.SUM_BY_NONZERO_WEIGHTS): pred/target (b,p,c), weights (b,p) broadcast over c.
 * sums[b] = sum huber(pred-target)*w; counts[b] = c * #(w != 0).  loss = sum(sums)/sum(counts) (0 if none). */
int mpsr_huber_loss_sums(const float *pred, const float *target, const float *weights, int b, int p, int c,
                         float delta, float *sums, float *counts, mpsr_stream_t stream);

/* grad (b,p,c) = scale[0] * w * clamp(pred - target, -delta, delta); scale is a DEVICE scalar (so the caller can
 * derive it from sums/counts without a host sync). */
int mpsr_huber_loss_grad(const float *pred, const float *target, const float *weights, const float *scale, int b,
                         int p, int c, float delta, float *grad, mpsr_stream_t stream);

/* monopsr_model.py:960-1071 format_predictions (test mode; lwh offset, alpha 'dc', view_ang est, centroids xyz)
 * with instance_utils.py:988-1032 postprocess_cen_x and monopsr_output_builder.py:805-860 score_boxes, one thread
 * per box in fp64: box_3d (b,9) = [x,y,z,l,w,h,ry,score,class-1], box_2d (b,7) = [y1,x1,y2,x2,alpha,score,class-1].
 * lwh (b,3); view_angs (b); alpha_bins/alpha_regs (b,num_alpha_bins); centroids (b,3); boxes_2d (b,4);
 * scores (b); class_idx (b) int32; cam_p (12). */
int mpsr_format_boxes(const float *lwh, const float *view_angs, const float *alpha_bins, const float *alpha_regs,
                      const float *centroids, const float *boxes_2d, const float *scores, const int *class_idx,
                      const float *cam_p, int b, int num_alpha_bins, int img_h, int img_w, int centroid_middle,
                      int post_process_cen_x, float max_depth, float *box_3d, float *box_2d, mpsr_stream_t stream);

/* ------------------------------------------------------------------------------------------------ network */

/* One packed convolution / FC layer inside a weight blob (offsets in floats from the blob base). */
typedef struct mpsr_layer {
    int32_t cin, cout, kh, kw, dilation, relu;
    int64_t w_off; /* (cout, kh*kw*cin) BN-folded weights */
    int64_t b_off; /* (cout) folded bias, or -1 */
} mpsr_layer;

/* Optional arguments of the *_ex network entry points (NULL = none; zero-initialise, then set what is wanted).
 *  - filter_cache: the Winograd kernels (F(3x3,3x3) on block3's atrous layers, F(4x4,3x3) on the map decoder) multiply
 *    by TRANSFORMED filters G g G^T.  Without a cache they are recomputed from `blob` by every call (a launch or a tail
 *    job per layer).  With a caller-owned buffer of mpsr_filter_cache_floats() floats they are written there by the
 *    first call (filter_cache_valid = 0) and only read by later calls that pass filter_cache_valid != 0 -- the caller
 *    vouches that blob / layers / B / shape are the ones the cache was filled for (inference: weights never change;
 *    a training loop passes 0 or no cache).  A cache that is too small is used for the layers that fit.  Which form a
 *    layer's filters take depends on the kernel the call picks (batch size, mpsr_set_conv_math,
 *    mpsr_set_winograd_policy): pass filter_cache_tags and the library keeps track itself.
 *  - ready_event (mpsr_squash_decoder_fwd_ex): recorded on `stream` as soon as feat_box3d is complete, so that the
 *    caller can start the FC heads (mpsr_heads_fwd) on ANOTHER stream while the map decoder still runs -- the heads
 *    are many small launches that fill the decoder kernels' partially occupied last rounds (reference graph: the two
 *    branches of net_builder.py:68-89 / monopsr_output_builder.py:126-194 are independent). */
typedef struct mpsr_net_opts {
    float *filter_cache;
    size_t filter_cache_floats;
    int32_t filter_cache_valid;
    mpsr_event_t ready_event;
    /* optional HOST array, one int per layer record (n_layers of the call), zero-initialised by the caller and kept with
     * the cache: the library notes in it WHAT it stored in a layer's slice (F(4x4,3x3) / F(3x3,3x3) transformed filters,
     * the tap GEMM's re-ordered rows) and re-fills a slice whose note does not match what the call is about to read --
     * so a cache survives a change of arithmetic mode, Winograd policy or batch size without the caller tracking it.
     * NULL: filter_cache_valid alone decides. */
    int32_t *filter_cache_tags;
    /* ABI 5: arithmetic mode / Winograd policy of this call (MPSR_CALL_MATH_*, MPSR_CALL_WINOGRAD_*; 0 = the process-wide
     * defaults).  With filter_cache_tags the cache follows a change between calls by itself. */
    int32_t math;
    int32_t winograd_policy;
} mpsr_net_opts;

/* Floats of filter cache that serve every 3x3 layer of `layers` (36 cout cin per dense layer, 25 per atrous one). */
size_t mpsr_filter_cache_floats(const mpsr_layer *layers, int n_layers);

#define MPSR_TRUNK_LAYERS 94 /* root + 30 bottleneck units x 3 + 3 projection shortcuts */

/* Scratch bytes for mpsr_trunk_fwd on a (B,H,W,3) input. */
size_t mpsr_trunk_workspace_bytes(int B, int H, int W);

/* ResNet-101 v1 to block3 at output stride 4 (rates 1/2/4), frozen BatchNorm folded, as
 * FasterRCNNResnet101FeatureExtractor._extract_proposal_features builds it
 * (faster_rcnn_resnet_v1_feature_extractor.py:197-245; resnet_v1.py:79-139,221-236,310-330;
 * resnet_utils.py:176-219).  img (B,H,W,3) -> out (B,H/4,W/4,1024).  layers[] in execution order:
 * root(as 1x1 over im2col, cin=160), then per unit [shortcut?], conv1, conv2, conv3. */
int mpsr_trunk_fwd(const float *img, int B, int H, int W, const float *blob, const mpsr_layer *layers,
                   int n_layers, float *out, void *workspace, size_t workspace_bytes, mpsr_stream_t stream);
/* The same with options (filter cache; ready_event is ignored). */
int mpsr_trunk_fwd_ex(const float *img, int B, int H, int W, const float *blob, const mpsr_layer *layers,
                      int n_layers, float *out, void *workspace, size_t workspace_bytes, const mpsr_net_opts *opts,
                      mpsr_stream_t stream);

#define MPSR_DECODER_LAYERS 7 /* squash 1x1 as two K-halves (crop, full), conv2 x2, conv3 x2, xyz 3x3 */

size_t mpsr_decoder_workspace_bytes(int B, int fh, int fw, int mh, int mw);

/* net_builder.py:62-89 + monopsr_output_builder.py:95-104: concat(crop_feat, full_feat) -> 1x1 conv 512 + ReLU
 * (features_squashed) -> 2x2 max-pool (features_for_box_3d) ; bilinear to (mh/2,mw/2) -> 2x[3x3 conv 256,BN,ReLU]
 * -> bilinear to (mh,mw) -> 2x[3x3 conv 128,BN,ReLU] (features_for_map) -> 3x3 conv 3 (inst_xyz_map_local).
 * crop_feat, full_feat (B,fh,fw,1024).  Outputs: feat_box3d (B,fh/2,fw/2,512), feat_map (B,mh,mw,128) or NULL to
 * skip storing it separately, xyz_map (B,mh,mw,3). */
int mpsr_squash_decoder_fwd(const float *crop_feat, const float *full_feat, int B, int fh, int fw, int mh, int mw,
                            const float *blob, const mpsr_layer *layers, int n_layers, float *feat_box3d,
                            float *feat_map, float *xyz_map, void *workspace, size_t workspace_bytes,
                            mpsr_stream_t stream);
/* Which kernel serves each of the seven layers for this shape (kinds[7], as mpsr_conv2d_plan; 7 = tap GEMM on the
 * source map + gather, csrc/upconv.hip) and the multiply-add FLOPs it issues (executed_flops[7]): throughput accounting
 * for bench.py's roofline object, not needed to run anything. */
int mpsr_squash_decoder_plan(int B, int fh, int fw, int mh, int mw, const mpsr_layer *layers, int n_layers, int *kinds,
                             double *executed_flops);
/* The same with options (filter cache; ready_event = feat_box3d complete). */
int mpsr_squash_decoder_fwd_ex(const float *crop_feat, const float *full_feat, int B, int fh, int fw, int mh, int mw,
                               const float *blob, const mpsr_layer *layers, int n_layers, float *feat_box3d,
                               float *feat_map, float *xyz_map, void *workspace, size_t workspace_bytes,
                               const mpsr_net_opts *opts, mpsr_stream_t stream);

#define MPSR_HEAD_LAYERS 7 /* img_fc (both heads fused along N), prop fc0, fc1, lwh+alpha, reg fc0, fc1, cen_y+cen_z */

typedef struct mpsr_head_consts {
    float image_h, image_w;   /* model_config.image_input_shape (320, 1216) */
    float max_depth;          /* dataset depth_range[1] (45) */
    float cen_y_norm;         /* 1.666754, monopsr_output_builder.py:246 */
    float cen_y_class_offset; /* 0.0648 for Car/kitti, instance_utils.py:933-937 */
    int32_t num_classes;      /* len(dataset_config.classes) */
    int32_t num_alpha_bins;   /* 12 */
} mpsr_head_consts;

size_t mpsr_heads_workspace_bytes(int B, int feat_elems);

/* monopsr_output_builder.py:126-302,407-438,457-488,509-661 wired as monopsr_model.py:320-413 with
 * output_config {lwh: offset, alpha: dc, view_ang: est, cen_x: from_view_ang_and_z, cen_y: offset, cen_z: offset}.
 *   feat_box3d (B,feat_elems) flattened NHWC features_for_box_3d
 *   boxes_2d (B,4) [y1,x1,y2,x2] pixels; cam_p (12) row-major 3x4; view_angs (B); class_idx (B) int32;
 *   mean_lwh (B,3); cen_z_offset (B)
 * Outputs (any may be NULL): lwh (B,3), lwh_offs (B,3), alpha_bins (B,nb), alpha_regs (B,nb), prop_cen_z (B),
 * cen_y (B), cen_y_offs (B), cen_z (B), cen_z_offs (B), cen_x (B), centroids (B,3). */
typedef struct mpsr_head_outputs {
    float *lwh, *lwh_offs, *alpha_bins, *alpha_regs, *prop_cen_z, *cen_y, *cen_y_offs, *cen_z, *cen_z_offs,
        *cen_x, *centroids;
} mpsr_head_outputs;

int mpsr_heads_fwd(const float *feat_box3d, int B, int feat_elems, const float *boxes_2d, const float *cam_p,
                   const float *view_angs, const int *class_idx, const float *mean_lwh, const float *cen_z_offset,
                   const mpsr_head_consts *consts, const float *blob, const mpsr_layer *layers, int n_layers,
                   const mpsr_head_outputs *outs, void *workspace, size_t workspace_bytes, mpsr_stream_t stream);
/* ABI 5: the boxes of several images in one call -- cam_p (n_cams,12), cam_index (B) int32 in [0, n_cams) picks each
 * box's projection matrix (NULL: every box uses the first; an index outside the range is never dereferenced -- the
 * outputs of that box that depend on the projection come back NaN).  The reference feeds one image per step
 * (monopsr_model.py:95-99, one pl_cam_p); N images x 32 boxes in one call read the 150 MB of FC weights once. */
int mpsr_heads_fwd_cams(const float *feat_box3d, int B, int feat_elems, const float *boxes_2d, const float *cam_p,
                        int n_cams, const int *cam_index, const float *view_angs, const int *class_idx,
                        const float *mean_lwh, const float *cen_z_offset, const mpsr_head_consts *consts,
                        const float *blob, const mpsr_layer *layers, int n_layers, const mpsr_head_outputs *outs,
                        void *workspace, size_t workspace_bytes, mpsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MONOPSR_HIP_H */
