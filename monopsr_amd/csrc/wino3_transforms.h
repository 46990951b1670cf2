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

}  // namespace w3t
}  // namespace mpsr
