// The filter transform of the F(3x3,3x3) atrous kernel (winograd3.hip) as a device function: winograd3.hip's own
// launch uses it, and the persistent pointwise kernel (pointwise.hip) runs it as a tail job -- the filters of the 3x3
// layer that FOLLOWS a 1x1 layer are transformed by that layer's workgroups when they have finished their tiles,
// instead of by a 6-us launch of their own in front of every one of block3's 23 atrous layers.
#pragma once
#include <hip/hip_runtime.h>

namespace mpsr {
namespace f3f {
constexpr int KC = 8, NP = 25;
}

// U[cb][pos][n][8] = (G' g G'^T)[pos], pos = 5 u + v, for filter g = w[n][(ky*3+kx)*C + c], c = cb*8 + j; i = n * C + c.
__device__ __forceinline__ void wino3_filter_one(const float *__restrict__ w, int N, int C, float *__restrict__ u, long long i)
{
    const int n = (int)(i / C), c = (int)(i - (long long)n * C);
    double g[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = (double)w[(size_t)n * 9 * C + (size_t)(ky * 3 + kx) * C + c];
    auto gcol = [](double a, double b, double c2, double *o) {  // G' (5x3) applied to one 3-vector
        o[0] = a / 2.0;
        o[1] = (a + b + c2) / 2.0;
        o[2] = (a - b + c2) / 6.0;
        o[3] = a / 6.0 + b / 3.0 + c2 * (2.0 / 3.0);
        o[4] = c2;
    };
    double t[5][3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        double o[5];
        gcol(g[0][kx], g[1][kx], g[2][kx], o);
#pragma unroll
        for (int r = 0; r < 5; ++r) t[r][kx] = o[r];
    }
    float *dst = u + ((size_t)(c / f3f::KC) * f3f::NP * N + n) * f3f::KC + (c % f3f::KC);
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        double o[5];
        gcol(t[r][0], t[r][1], t[r][2], o);
#pragma unroll
        for (int s = 0; s < 5; ++s) dst[(size_t)(r * 5 + s) * N * f3f::KC] = (float)o[s];
    }
}

// The same for the sixteen-product form of a zero-padded 3x3 tile (wino3_transforms.h, winograd3z.hip):
// U[cb][pos][n][8] = (G g G^T)[pos], pos = 4 u + v, G = [-1/4 1/8 1/4; 1/4 1/8 -1/4; 1/6 1/4 1/6; -1/6 1/4 -1/6].
namespace f3z {
constexpr int KC = 8, NP = 16;
}
__device__ __forceinline__ void wino3z_filter_one(const float *__restrict__ w, int N, int C, float *__restrict__ u, long long i)
{
    const int n = (int)(i / C), c = (int)(i - (long long)n * C);
    double g[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = (double)w[(size_t)n * 9 * C + (size_t)(ky * 3 + kx) * C + c];
    auto gcol = [](double a, double b, double c2, double *o) {  // G (4x3) applied to one 3-vector (a, b, c2) = (g0, g1, g2)
        const double u = (c2 - a) / 4.0, s = (a + c2) / 6.0;
        o[0] = b / 8.0 + u;
        o[1] = b / 8.0 - u;
        o[2] = b / 4.0 + s;
        o[3] = b / 4.0 - s;
    };
    double t[4][3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        double o[4];
        gcol(g[0][kx], g[1][kx], g[2][kx], o);
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r][kx] = o[r];
    }
    float *dst = u + ((size_t)(c / f3z::KC) * f3z::NP * N + n) * f3z::KC + (c % f3z::KC);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double o[4];
        gcol(t[r][0], t[r][1], t[r][2], o);
#pragma unroll
        for (int s = 0; s < 4; ++s) dst[(size_t)(r * 4 + s) * N * f3z::KC] = (float)o[s];
    }
}

// A filter-transform job handed from a network-level entry point (network.hip) to the next pointwise launch on this
// thread, and the note that it was done, for conv3x3_winograd3 to find.
struct FilterTailJob {
    const float *w = nullptr;
    float *u = nullptr;
    int N = 0, C = 0;
    int form = 0;  // 0: F(3x3,3x3), 25 positions (wino3_filter_one); 1: the sixteen-product form (wino3z_filter_one)
};
extern thread_local FilterTailJob g_filter_tail_job;    // pending: consumed by conv1x1_pointwise
extern thread_local FilterTailJob g_filter_tail_done;   // done by the last pointwise launch: consumed by conv3x3_winograd3

// Where the transformed filters of the layer about to run live when the caller keeps them across calls
// (mpsr_net_opts.filter_cache): set by the network entry points (network.hip) right before the layer, consumed by
// conv3x3_winograd3 / conv3x3_winograd4, which then use `u` instead of their scratch and skip the transform if `ready`.
struct FilterCacheSlot {
    const float *w = nullptr;
    float *u = nullptr;
    size_t floats = 0;
    bool ready = false;
    int *tag = nullptr;  // caller's note of what the slice holds (mpsr_net_opts.filter_cache_tags), or nullptr
    // true when the slice already holds form `kind` of this layer's filters; notes `kind` for the next call either way
    // (the consumer is about to write it if not)
    bool holds(int kind)
    {
        const bool ok = ready && (!tag || *tag == kind);
        if (tag) *tag = kind;
        return ok;
    }
};
enum { FILTER_FORM_WINO4 = 1, FILTER_FORM_WINO3 = 2, FILTER_FORM_UPCONV = 3, FILTER_FORM_WINO3Z = 4, FILTER_FORM_WINO2 = 5 };
extern thread_local FilterCacheSlot g_filter_cache_slot;
}  // namespace mpsr
