// The 1-D transforms of Winograd F(3x3,3x3) on points 0, 1, -1, 2, inf with rows of B^T / G scaled by (2, 2, 6, 6, 1)
// (integer constants in the input transform), shared by the two F(3x3,3x3) kernels: winograd3.hip (eight waves share a
// tile's 25 positions) and winograd3w.hip (one wave owns all 25).  The filter side is wino3_filter.h.
#pragma once
#include <hip/hip_runtime.h>

namespace mpsr {
namespace w3t {

// scaled B^T applied to (0, x0, x1, x2, 0): 7 operations
//   t0 = -x0 - 2 x1 + x2,  t1 = 2 x0 + x1 - x2,  t2 = -2 x0 + 3 x1 - x2,  t3 = -x0 + x2,  t4 = 2 x0 - x1 - 2 x2
__device__ __forceinline__ void bt5(float x0, float x1, float x2, float &t0, float &t1, float &t2, float &t3, float &t4)
{
    const float e = x2 - x0, f = x0 - e, g = fmaf(-2.f, x0, -x2);
    t0 = fmaf(-2.f, x1, e);
    t1 = x1 + f;
    t2 = fmaf(3.f, x1, g);
    t3 = e;
    t4 = fmaf(-2.f, e, -x1);
}
// A^T (3x5) applied to a 5-vector: 7 operations
__device__ __forceinline__ void at3(float m0, float m1, float m2, float m3, float m4, float &y0, float &y1, float &y2)
{
    const float p = m1 + m2, q = m1 - m2;
    y0 = m0 + p + m3;
    y1 = fmaf(2.f, m3, q);
    y2 = fmaf(4.f, m3, p) + m4;
}


// ---- The zero-padded 3x3 tile in SIXTEEN products (r05).
// A pixel sub-grid that is exactly one 3x3 tile (H = W = 3 x dilation) reads nothing outside itself: in one dimension its
// three outputs are y_k = sum_a g_a d_(k+a-1) with d_(-1) = d_3 = 0 -- the MIDDLE three coefficients of the product of two
// quadratics, (c1, c2, c3) of (g2 + g1 x + g0 x^2)(d0 + d1 x + d2 x^2).  General F(3,3) needs 5 multiplications because
// it serves any 5-point input; this bilinear map has rank 4 (the structure tensor is the fully symmetric "index sum = 3"
// tensor; over the complex numbers the four products are evaluations at the 4th roots of unity, over the reals):
//     u = g2 - g0, s = g2 + g0, U = d0 - d2, S = d0 + d2
//     P1 = (g1 + 2 u)(d1 + 2 U), P2 = (g1 - 2 u)(d1 - 2 U), P3 = (g1 + 2/3 s)(d1 + 2/3 S), P4 = (g1 - 2/3 s)(d1 - 2/3 S)
//     y0 + y2 = 3/4 (P3 - P4),  y0 - y2 = 1/4 (P1 - P2),  y1 = 9/16 (P3 + P4) - 1/16 (P1 + P2)
// (check: P1 - P2 = 4 (g1 U + u d1), P3 - P4 = 4/3 (g1 S + s d1), s S - u U = 2 (g0 d2... + g2 d0), 1 / z^2 - 1 / x^2 = 2 for
// the two evaluation scales x = 2, z = 2/3).  Nested in two dimensions: 16 products per (channel pair, tile) where
// F(3x3,3x3) has 25 and the direct form 81 -- and with constants 2, 3/2, 1/2 instead of F(3,3)'s 2, 3, 4, 1/6 the fp32 error
// is that of a direct convolution: 5e-7 of the tensor scale, 1.8e-4 element-wise on the heavy-tailed maps of
// tests/test_hostile_inputs_gpu.py, where F(3x3,3x3) has 4e-6 and 1.2e-3.  With the scales folded into the filter side:
//     B^T (4x3) = [2 1 -2; -2 1 2; 1 3/2 1; -1 3/2 -1],   G (4x3) = [-1/4 1/8 1/4; 1/4 1/8 -1/4; 1/6 1/4 1/6; -1/6 1/4 -1/6],
//     A^T (3x4) = [1 -1 1 -1; -1/2 -1/2 3/2 3/2; -1 1 1 -1].
// input side, 6 operations: (x0, x1, x2) -> B^T x
__device__ __forceinline__ void bt4z(float x0, float x1, float x2, float &t0, float &t1, float &t2, float &t3)
{
    const float U = x0 - x2, S = x0 + x2;
    t0 = fmaf(2.f, U, x1);
    t1 = fmaf(-2.f, U, x1);
    t2 = fmaf(1.5f, x1, S);
    t3 = fmaf(1.5f, x1, -S);
}
// output side, 8 operations: (m0..m3) -> A^T m
__device__ __forceinline__ void at3z(float m0, float m1, float m2, float m3, float &y0, float &y1, float &y2)
{
    const float d01 = m0 - m1, s01 = m0 + m1, d23 = m2 - m3, s23 = m2 + m3;
    y0 = d23 + d01;
    y1 = fmaf(1.5f, s23, -0.5f * s01);
    y2 = d23 - d01;
}
// gradient side, 6 operations: (d0, d1, d2) -> A d (A = the transpose of A^T: 4x3)
__device__ __forceinline__ void a4z(float d0, float d1, float d2, float &o0, float &o1, float &o2, float &o3)
{
    const float U = d0 - d2, S = d0 + d2;
    o0 = fmaf(-0.5f, d1, U);
    o1 = fmaf(-0.5f, d1, -U);
    o2 = fmaf(1.5f, d1, S);
    o3 = fmaf(1.5f, d1, -S);
}

}  // namespace w3t
}  // namespace mpsr
