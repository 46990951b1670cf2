// Shared host-side helpers for libmonopsr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "monopsr_hip.h"

namespace mpsr {

// Thread-local last-error buffer behind mpsr_last_error().
char *error_buffer();
constexpr int kErrorBufferBytes = 512;

inline int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), kErrorBufferBytes, fmt, ap);
    va_end(ap);
    return code;
}

inline hipStream_t as_stream(mpsr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Launch check: kernel launches report configuration errors through hipGetLastError().
#define MPSR_CHECK_LAUNCH(what)                                                                       \
    do {                                                                                              \
        hipError_t e_ = hipGetLastError();                                                            \
        if (e_ != hipSuccess) return ::mpsr::fail(MPSR_ERR_HIP, "%s: %s", what, hipGetErrorString(e_)); \
    } while (0)

#define MPSR_CHECK_HIP(expr)                                                                          \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return ::mpsr::fail(MPSR_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define MPSR_REQUIRE(cond, ...)                                                    \
    do {                                                                           \
        if (!(cond)) return ::mpsr::fail(MPSR_ERR_INVALID_ARG, __VA_ARGS__);       \
    } while (0)

// Per-call overrides of the process-wide arithmetic mode / Winograd policy (mpsr_net_opts.math / .winograd_policy,
// mpsr_conv2d_nhwc_f32_ex): an entry point runs on its caller's thread from start to end, so a thread-local set for
// the duration of the call IS per call, and two threads with different options never see each other's.  -1 = none.
extern thread_local int t_call_math;
extern thread_local int t_call_wino_policy;
struct CallOptsGuard {  // sets the overrides for one entry-point call (0 = inherit, otherwise enum value + 1)
    int prev_math, prev_wino;
    CallOptsGuard(int math_opt, int wino_opt) : prev_math(t_call_math), prev_wino(t_call_wino_policy)
    {
        if (math_opt > 0) t_call_math = math_opt - 1;
        if (wino_opt > 0) t_call_wino_policy = wino_opt - 1;
    }
    ~CallOptsGuard()
    {
        t_call_math = prev_math;
        t_call_wino_policy = prev_wino;
    }
};

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }


// Division of a non-negative int (< 2^31) by a launch constant as multiply-high + shift (Granlund-Montgomery,
// m = ceil(2^(31+l) / d), l = ceil(log2 d)).  On this hardware every vector-ALU instruction of a resident wave takes
// ~4 cycles away from the SIMD's matrix pipe (tools/mfma_peak.py --valu), and hipcc's integer division is a ~35
// instruction sequence: the row -> (image, y, x) decodes alone cost the K = 256 layers a fifth of their MFMA time.
struct FastDiv {
    unsigned m;  // 0: divisor 1
    int s;
};
__host__ __device__ inline FastDiv make_fastdiv(int d)
{
    FastDiv f{0u, 0};
    if (d <= 1) return f;
    int l = 0;
    while ((1LL << l) < d) ++l;
    f.m = (unsigned)((((unsigned long long)1 << (31 + l)) + (unsigned)d - 1) / (unsigned)d);
    f.s = l - 1;
    return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv &f)
{
    return f.m == 0 ? n : (int)(__umulhi((unsigned)n, f.m) >> f.s);
}

}  // namespace mpsr
