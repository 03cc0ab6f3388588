// wfs_common.h -- internal helpers of libwfsparse (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/wfsparse.h"

#define WFS_WAVE 64

void wfs_set_error(const char *fmt, ...);

#define WFS_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            wfs_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                          __LINE__);                                                         \
            return WFS_EHIP;                                                                 \
        }                                                                                    \
    } while (0)

#define WFS_LAUNCH_CHECK() WFS_HIP_CHECK(hipGetLastError())

#define WFS_REQUIRE(cond, code, ...)                                                         \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            wfs_set_error(__VA_ARGS__);                                                      \
            return (code);                                                                   \
        }                                                                                    \
    } while (0)

static inline size_t wfs_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int64_t wfs_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// bf16 storage helpers (round-to-nearest-even through the hardware cvt; NaN stays NaN)
typedef unsigned short wfs_bf16;
__device__ __forceinline__ float wfs_ld(const float *p) { return *p; }
__device__ __forceinline__ float wfs_ld(const wfs_bf16 *p) {
    return __uint_as_float(((unsigned)*p) << 16);
}
__device__ __forceinline__ void wfs_st(float *p, float v) { *p = v; }
__device__ __forceinline__ void wfs_st(wfs_bf16 *p, float v) {
    __bf16 b = (__bf16)v;
    *p = *reinterpret_cast<unsigned short *>(&b);
}

// event timing (opt-in; see wfs_timing_enable)
struct WfsTimerScope {
    int timer;
    hipStream_t stream;
    void *rec;
    WfsTimerScope(int timer, hipStream_t stream);
    ~WfsTimerScope();
};

// shape-specialised launchers (conv_mfma.hip); r_dev: optional device-side count of valid rows (<= R)
bool wfs_mfma_gconv32_ok(int K);
// stats (optional): BatchNorm statistics taken in the epilogue (conv_stats.h); forward products only
int wfs_launch_gconv32_f32(const int *table, int mirror, int K, int identity_k, long long R, const long long *r_dev,
                           const float *X, const float *W, int transpose_w, const float *bias, float *Y,
                           const wfs_bn_stats *stats, hipStream_t stream);
int wfs_launch_gconv32_bf16(const int *table, int mirror, int K, int identity_k, long long R, const long long *r_dev,
                            const void *X, const float *W, int transpose_w, const float *bias, void *Y,
                            const wfs_bn_stats *stats, hipStream_t stream);
int wfs_launch_gconv_c2c32(const int *table, const int *kmap, int K, int identity_k, long long R,
                           const long long *r_dev, const void *X, const float *W, const float *bias, void *Y, int dtype,
                           const wfs_bn_stats *stats, bool *stats_done, hipStream_t stream);
size_t wfs_conv_stats_fast_workspace(long long R);
// bn.hip: the stand-alone statistics pass (reduce + fold) over X [N, C]
int wfs_launch_bn_stats(const void *X, long long N, int C, int dtype, const long long *n_dev, const wfs_bn_stats *stats,
                        hipStream_t stream);
size_t wfs_dw_fast_workspace(int K, long long R, int Cs, int Cg);
int wfs_launch_gdw32(const int *table, int K, int identity_k, long long R, const long long *r_dev, const void *S,
                     const void *G, int swap, float *dW, float *part, int dtype, hipStream_t stream);
int wfs_launch_gdw_c32c2(const int *table, int mirror, int K, int identity_k, long long R, const long long *r_dev,
                         const void *S, const void *G, int swap, float *dW, float *part, int dtype,
                         hipStream_t stream);
