// The five launcher functions the reference's TensorFlow op shells call
// (tf_ops/nn_distance/tf_nndistance.cpp:168,208; tf_ops/approxmatch/tf_approxmatch.cpp:141-143), implemented on
// libmonopsr_hip.so.  Same names, same C++ signatures, same (void) return: compile this file instead of the
// reference's tf_nndistance_g.cu / tf_approxmatch_g.cu and link -lmonopsr_hip; the op shells stay untouched.
//
//   g++ -std=c++11 -fPIC -I include -c examples/reference_op_launchers.cpp
//   g++ -shared -o libmonopsr_tf_launchers.so reference_op_launchers.o -Lmonopsr_amd -lmonopsr_hip
//
// The launchers run on the null stream like the reference's kernels (tf_nndistance_g.cu:129); with a ROCm TensorFlow
// pass ctx->eigen_device<GPUDevice>().stream() instead (MPSR_LAUNCHER_STREAM).  tests/test_reference_launchers.py
// compiles and links this file (CPU) and drives it with device buffers (GPU).
#include <cstdio>
#include <cstdlib>

#include "monopsr_hip.h"

#ifndef MPSR_LAUNCHER_STREAM
#define MPSR_LAUNCHER_STREAM nullptr
#endif

// The reference's launchers cannot report errors; its shells validated the shapes already (OP_REQUIRES), so a
// non-zero status here is a programming or device error: say why and stop, as LOG(FATAL) would.
static void check(int status, const char *what)
{
    if (status == MPSR_OK) return;
    std::fprintf(stderr, "%s: %s\n", what, mpsr_last_error());
    std::abort();
}

void NmDistanceKernelLauncher(int b, int n, const float *xyz, int m, const float *xyz2, float *result, int *result_i,
                              float *result2, int *result2_i)
{
    check(mpsr_nn_distance_fwd(b, n, xyz, m, xyz2, result, result_i, result2, result2_i, MPSR_LAUNCHER_STREAM),
          "NmDistanceKernelLauncher");
}

void NmDistanceGradKernelLauncher(int b, int n, const float *xyz1, int m, const float *xyz2, const float *grad_dist1,
                                  const int *idx1, const float *grad_dist2, const int *idx2, float *grad_xyz1,
                                  float *grad_xyz2)
{
    check(mpsr_nn_distance_bwd(b, n, xyz1, m, xyz2, grad_dist1, idx1, grad_dist2, idx2, grad_xyz1, grad_xyz2,
                               MPSR_LAUNCHER_STREAM),
          "NmDistanceGradKernelLauncher");
}

// `temp` is what the reference's shell allocates: TensorShape{b, (n+m)*2} floats (tf_approxmatch.cpp:167-168).  That
// size makes the library keep its per-level state in the tail of `match` itself until those rows are written (same
// result bit for bit and the same speed as with mpsr_approx_match_temp_floats(b, n, m) floats of scratch).
void approxmatchLauncher(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp)
{
    check(mpsr_approx_match(b, n, m, xyz1, xyz2, match, temp, (size_t)b * (size_t)(n + m) * 2, MPSR_LAUNCHER_STREAM),
          "approxmatchLauncher");
}

void matchcostLauncher(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *out)
{
    check(mpsr_match_cost(b, n, m, xyz1, xyz2, match, out, MPSR_LAUNCHER_STREAM), "matchcostLauncher");
}

void matchcostgradLauncher(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *grad1,
                           float *grad2)
{
    check(mpsr_match_cost_grad(b, n, m, xyz1, xyz2, match, grad1, grad2, MPSR_LAUNCHER_STREAM), "matchcostgradLauncher");
}
