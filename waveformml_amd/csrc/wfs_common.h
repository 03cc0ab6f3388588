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

// 16-bit storage helpers (round-to-nearest-even through the hardware cvt; NaN stays NaN).  bf16 rows are plain
// unsigned shorts; fp16 rows get a distinct type so that overloads and templates can tell the two apart.
typedef unsigned short wfs_bf16;
struct wfs_f16 {
    unsigned short bits;
};
__device__ __forceinline__ float wfs_ld(const float *p) { return *p; }
__device__ __forceinline__ float wfs_ld(const wfs_bf16 *p) {
    return __uint_as_float(((unsigned)*p) << 16);
}
__device__ __forceinline__ float wfs_ld(const wfs_f16 *p) {
    return (float)__builtin_bit_cast(_Float16, p->bits);
}
__device__ __forceinline__ void wfs_st(float *p, float v) { *p = v; }
__device__ __forceinline__ void wfs_st(wfs_bf16 *p, float v) {
    __bf16 b = (__bf16)v;
    *p = *reinterpret_cast<unsigned short *>(&b);
}
__device__ __forceinline__ void wfs_st(wfs_f16 *p, float v) {
    p->bits = __builtin_bit_cast(unsigned short, (_Float16)v);
}
// one dword = two neighbouring 16-bit elements (low half first)
template <typename H>
__device__ __forceinline__ void wfs_unpack2(unsigned w, float &lo, float &hi);
template <>
__device__ __forceinline__ void wfs_unpack2<wfs_bf16>(unsigned w, float &lo, float &hi) {
    lo = __uint_as_float(w << 16);
    hi = __uint_as_float(w & 0xFFFF0000u);
}
template <>
__device__ __forceinline__ void wfs_unpack2<wfs_f16>(unsigned w, float &lo, float &hi) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 v = __builtin_bit_cast(h2, w);
    lo = (float)v.x;
    hi = (float)v.y;
}
template <typename H>
__device__ __forceinline__ unsigned wfs_pack2(float lo, float hi);
template <>
__device__ __forceinline__ unsigned wfs_pack2<wfs_bf16>(float lo, float hi) {
    wfs_bf16 a, b;
    wfs_st(&a, lo);
    wfs_st(&b, hi);
    return (unsigned)a | ((unsigned)b << 16);
}
template <>
__device__ __forceinline__ unsigned wfs_pack2<wfs_f16>(float lo, float hi) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(unsigned, v);
}
// the value a float takes when it is stored in H and read back
template <typename H>
__device__ __forceinline__ float wfs_round_to(float v) {
    H t;
    wfs_st(&t, v);
    return wfs_ld(&t);
}
static inline bool wfs_dtype_ok(int dtype) { return dtype == WFS_F32 || dtype == WFS_BF16 || dtype == WFS_F16; }

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
int wfs_launch_gconv32_f32(const int *table, int mirror, int K, int identity_k, long long R, const long long *r_dev,
                           const float *X, const float *W, int transpose_w, const float *bias, float *Y,
                           hipStream_t stream, int packed_kl = 0);
// 16-bit rows (dtype WFS_BF16 or WFS_F16)
int wfs_launch_gconv32_h16(const int *table, int mirror, int K, int identity_k, long long R, const long long *r_dev,
                           const void *X, const float *W, int transpose_w, const float *bias, void *Y, int dtype,
                           hipStream_t stream, int packed_kl = 0);
int wfs_launch_gconv_c2c32(const int *table, const int *kmap, int K, int identity_k, long long R,
                           const long long *r_dev, const void *X, const float *W, const float *bias, void *Y, int dtype,
                           hipStream_t stream);
// wide layers (wide.hip): dW as dense matrix-core products
bool wfs_wide_dw_ok(int K, long long R, int Cs, int Cg, int dtype);
size_t wfs_wide_dw_workspace(int K, long long R, int Cs, int Cg, int dtype);
int wfs_launch_wide_dw(const int *table, const int *kmap_host, int K, int identity_k, long long R, const long long *r_dev,
                       const void *S, int Cs, const void *G, long long G_rows, int Cg, int swap, float *dW, int dtype,
                       void *workspace, size_t workspace_bytes, hipStream_t stream);
size_t wfs_dw_fast_workspace(int K, long long R, int Cs, int Cg);
int wfs_launch_gdw32(const int *table, int K, int identity_k, long long R, const long long *r_dev, const void *S,
                     const void *G, int swap, float *dW, float *part, int dtype, wfs_dw_job *defer, hipStream_t stream,
                     int packed_kl = 0);
// dW and dX of a 32 -> 32 layer with 16-bit rows in ONE launch (conv_mfma.hip k_bwd32_bf16)
bool wfs_bwd32_fused_ok(int K, int packed_kl, int dtype);
int wfs_launch_bwd32_h16(const int *table, int packed_kl, int K, int identity_k, long long R, const long long *r_dev,
                         const void *S, const void *G, const float *W, void *dX, int swap, float *dW, float *part, int dtype,
                         wfs_dw_job *defer, hipStream_t stream);
int wfs_launch_gdw_c32c2(const int *table, int mirror, int K, int identity_k, long long R, const long long *r_dev,
                         const void *S, const void *G, int swap, float *dW, float *part, int dtype,
                         wfs_dw_job *defer, hipStream_t stream);
int wfs_launch_dw_jobs(const wfs_dw_job *jobs, int n, hipStream_t stream);
