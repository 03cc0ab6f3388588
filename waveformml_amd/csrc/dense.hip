// dense.hip -- SparseConvTensor.dense() (spconv.ToDense) and its backward.
//
// Replaces spconv 1.2.1's SparseConvTensor.dense() = scatter_nd + permute + contiguous
// (reference call sites src/models/SPConvBlocks.py:81,515, src/engineering/
// LitBase.py:138-146): out = zeros([B, *spatial, C]); out[idx] = features (ASSIGNMENT, last row
// wins on duplicate coordinates); returned channels-first.  Here the channels-first tensor is
// written directly (no [B,*spatial,C] intermediate, no permute copy).
#include "wfs_common.h"

namespace {
constexpr int TB = 256;

struct Shape {
    int ndim;
    int spatial[4];
    long long volume;
};

__device__ __forceinline__ long long cell_of(const Shape &s, const int *row, int &b) {
    b = row[0];
    long long pos = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d)
        if (d < s.ndim) pos = pos * s.spatial[d] + row[1 + d];
    return pos;
}

// M = capacity (grid); the number of valid rows comes from device memory when m_dev is given
__device__ __forceinline__ long long valid_rows(long long M, const long long *m_dev) {
    long long v = m_dev ? *m_dev : M;
    return v < M ? v : M;
}

__global__ void k_dense_winner(Shape s, const int *__restrict__ idx, long long M, const long long *m_dev,
                               int *__restrict__ winner) {
    long long m = (long long)blockIdx.x * TB + threadIdx.x;
    if (m >= valid_rows(M, m_dev)) return;
    int b;
    long long pos = cell_of(s, idx + m * (s.ndim + 1), b);
    atomicMax(&winner[(long long)b * s.volume + pos], (int)m);
}

template <typename T>
__global__ void k_to_dense(Shape s, const T *__restrict__ X, const int *__restrict__ idx, long long M,
                           const long long *m_dev, int C, const int *__restrict__ winner, T *__restrict__ Y) {
    long long e = (long long)blockIdx.x * TB + threadIdx.x;
    if (e >= valid_rows(M, m_dev) * C) return;
    long long m = e / C;
    int c = (int)(e % C);
    int b;
    long long pos = cell_of(s, idx + m * (s.ndim + 1), b);
    if (winner && winner[(long long)b * s.volume + pos] != (int)m) return;
    Y[((long long)b * C + c) * s.volume + pos] = X[e];
}

template <typename T>
__global__ void k_to_dense_bwd(Shape s, const T *__restrict__ dY, const int *__restrict__ idx, long long M,
                               const long long *m_dev, int C, T *__restrict__ dX) {
    long long e = (long long)blockIdx.x * TB + threadIdx.x;
    if (e >= valid_rows(M, m_dev) * C) return;
    long long m = e / C;
    int c = (int)(e % C);
    int b;
    long long pos = cell_of(s, idx + m * (s.ndim + 1), b);
    // torch's index_put_ backward (what spconv's dense() differentiates through) hands the cell's
    // gradient to EVERY row that addressed it, overwritten ones included -- so no winner mask here.
    dX[e] = dY[((long long)b * C + c) * s.volume + pos];
}

int make_shape(Shape *s, int ndim, const int32_t *spatial_host) {
    WFS_REQUIRE(ndim >= 1 && ndim <= WFS_MAX_DIM && spatial_host, WFS_EINVAL, "bad ndim/spatial");
    s->ndim = ndim;
    s->volume = 1;
    for (int i = 0; i < 4; ++i) {
        s->spatial[i] = i < ndim ? spatial_host[i] : 1;
        s->volume *= s->spatial[i];
    }
    return WFS_OK;
}
}  // namespace

// winner_ws: NULL when the coordinates are known to be unique; otherwise int32 [B * volume] scratch,
// which makes "last row wins" deterministic.
extern "C" int wfs_to_dense(const void *X, const int32_t *indices, int64_t M, int32_t ndim,
                            const int32_t *spatial_host, int32_t batch_size, int32_t C, void *Y, int32_t *winner_ws,
                            int32_t dtype, const int64_t *m_dev_, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const long long *m_dev = (const long long *)m_dev_;
    Shape s;
    int rc = make_shape(&s, ndim, spatial_host);
    if (rc) return rc;
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    if (M == 0 || C == 0) return WFS_OK;
    WFS_REQUIRE(X && indices && Y, WFS_EINVAL, "NULL device pointer");
    if (winner_ws) {
        WFS_HIP_CHECK(hipMemsetAsync(winner_ws, 0xFF, (size_t)batch_size * s.volume * 4, stream));
        k_dense_winner<<<dim3((unsigned)wfs_cdiv(M, TB)), dim3(TB), 0, stream>>>(s, indices, M, m_dev, winner_ws);
        WFS_LAUNCH_CHECK();
    }
    dim3 grid((unsigned)wfs_cdiv(M * C, TB)), block(TB);
    if (dtype == WFS_F32)
        k_to_dense<float><<<grid, block, 0, stream>>>(s, (const float *)X, indices, M, m_dev, C, winner_ws, (float *)Y);
    else                                          // bf16 and fp16 alike: 2-byte elements, copied as they are
        k_to_dense<wfs_bf16><<<grid, block, 0, stream>>>(s, (const wfs_bf16 *)X, indices, M, m_dev, C, winner_ws,
                                                         (wfs_bf16 *)Y);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

extern "C" int wfs_to_dense_bwd(const void *dY, const int32_t *indices, int64_t M, int32_t ndim,
                                const int32_t *spatial_host, int32_t batch_size, int32_t C, void *dX,
                                int32_t dtype, const int64_t *m_dev_, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const long long *m_dev = (const long long *)m_dev_;
    (void)batch_size;
    Shape s;
    int rc = make_shape(&s, ndim, spatial_host);
    if (rc) return rc;
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    if (M == 0 || C == 0) return WFS_OK;
    WFS_REQUIRE(dY && indices && dX, WFS_EINVAL, "NULL device pointer");
    dim3 grid((unsigned)wfs_cdiv(M * C, TB)), block(TB);
    if (dtype == WFS_F32)
        k_to_dense_bwd<float><<<grid, block, 0, stream>>>(s, (const float *)dY, indices, M, m_dev, C, (float *)dX);
    else
        k_to_dense_bwd<wfs_bf16><<<grid, block, 0, stream>>>(s, (const wfs_bf16 *)dY, indices, M, m_dev, C,
                                                             (wfs_bf16 *)dX);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}
