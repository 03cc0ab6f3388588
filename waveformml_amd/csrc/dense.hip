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

// ---- dense() through the cell -> row map a regular conv's rulebook build leaves behind (wfs_rulebook_cell_map).
// With the map every CELL knows its row (or that it has none), so a block owns a 64-cell stretch of one event and
// writes ALL of it -- values and zeros -- channel by channel as whole 128/256-byte runs: no memset of the dense
// tensor, no scattered 2-byte stores (the row-parallel kernel above writes each of a row's C values V cells apart).
// Tile in LDS as 32-bit words [c][w]: a word holds PACK = 4 / sizeof(T) neighbouring cells of one channel.
constexpr int DM_CELLS = 64;
template <typename T>
__device__ __forceinline__ int mapped_row(const unsigned *ticket, const int *slot_id, long long cell, long long Mv) {
    if (ticket[cell] == 0xFFFFFFFFu) return -1;
    const int id = slot_id[cell];
    return (id >= 0 && id < Mv) ? id : -1;
}


template <typename T>
__global__ void __launch_bounds__(TB) k_to_dense_mapped(const T *__restrict__ X, const unsigned *__restrict__ ticket,
                                                        const int *__restrict__ slot_id, long long M,
                                                        const long long *m_dev, long long V, int C, T *__restrict__ Y) {
    constexpr int PACK = 4 / (int)sizeof(T);              // cells per 32-bit word: 2 (16-bit rows) or 1 (fp32)
    constexpr int WORDS = DM_CELLS / PACK;                // words per channel and tile
    extern __shared__ unsigned sT[];                      // [C][WORDS + 1]
    __shared__ int sRow[DM_CELLS];
    const long long Mv = valid_rows(M, m_dev);
    const long long b = blockIdx.y, p0 = (long long)blockIdx.x * DM_CELLS;
    if (threadIdx.x < DM_CELLS) {
        const long long p = p0 + threadIdx.x;
        sRow[threadIdx.x] = p < V ? mapped_row<T>(ticket, slot_id, b * V + p, Mv) : -1;
    }
    __syncthreads();
    // gather: one item = (word w of the tile, group q of 4 channels): PACK rows x 4 channels
    const int groups = C / 4;
    for (int it = threadIdx.x; it < WORDS * groups; it += TB) {
        const int w = it / groups, q = it % groups;
        unsigned word[4] = {0u, 0u, 0u, 0u};
        if (Mv > 0) {                                       // (uniform) unconditional clamped loads + select, no branches
#pragma unroll
            for (int h = 0; h < PACK; ++h) {
                const int row = sRow[w * PACK + h];
                const bool ok = row >= 0;
                const T *src = X + (long long)(ok ? row : 0) * C + 4 * q;
                if constexpr (sizeof(T) == 4) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(src);
                    unsigned e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) word[j] = ok ? e[j] : 0u;
                } else {
                    const uint2 v = *reinterpret_cast<const uint2 *>(src);          // 4 x 16 bit
                    unsigned e[4] = {v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16};
#pragma unroll
                    for (int j = 0; j < 4; ++j) word[j] |= ok ? e[j] << (16 * h) : 0u;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) sT[(4 * q + j) * (WORDS + 1) + w] = word[j];
    }
    __syncthreads();
    // write out: channel c, word w -> cells p0 + w * PACK ... of Y[b][c][.]
    unsigned *Yw = reinterpret_cast<unsigned *>(Y);
    const long long live_words = (V - p0 < DM_CELLS ? V - p0 : DM_CELLS) / PACK;   // V % PACK == 0 (checked by the host)
    for (int it = threadIdx.x; it < C * WORDS; it += TB) {
        const int c = it / WORDS, w = it % WORDS;
        if (w < live_words) Yw[((b * C + c) * V + p0) / PACK + w] = sT[c * (WORDS + 1) + w];
    }
}

// backward: dX[row][c] = dY[b][c][cell of row]; the same tiles read channel by channel, rows written whole
template <typename T>
__global__ void __launch_bounds__(TB) k_to_dense_bwd_mapped(const T *__restrict__ dY, const unsigned *__restrict__ ticket,
                                                            const int *__restrict__ slot_id, long long M,
                                                            const long long *m_dev, long long V, int C,
                                                            T *__restrict__ dX) {
    constexpr int PACK = 4 / (int)sizeof(T);
    constexpr int WORDS = DM_CELLS / PACK;
    extern __shared__ unsigned sT[];
    __shared__ int sRow[DM_CELLS];
    const long long Mv = valid_rows(M, m_dev);
    const long long b = blockIdx.y, p0 = (long long)blockIdx.x * DM_CELLS;
    if (threadIdx.x < DM_CELLS) {
        const long long p = p0 + threadIdx.x;
        sRow[threadIdx.x] = p < V ? mapped_row<T>(ticket, slot_id, b * V + p, Mv) : -1;
    }
    const unsigned *Gw = reinterpret_cast<const unsigned *>(dY);
    const long long live_words = (V - p0 < DM_CELLS ? V - p0 : DM_CELLS) / PACK;
    for (int it = threadIdx.x; it < C * WORDS; it += TB) {
        const int c = it / WORDS, w = it % WORDS;
        sT[c * (WORDS + 1) + w] = w < live_words ? Gw[((b * C + c) * V + p0) / PACK + w] : 0u;
    }
    __syncthreads();
    const int groups = C / 4;
    for (int it = threadIdx.x; it < DM_CELLS * groups; it += TB) {
        const int cell = it / groups, q = it % groups;
        const int row = sRow[cell];
        if (row < 0) continue;
        const int w = cell / PACK, h = cell % PACK;
        unsigned e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = sT[(4 * q + j) * (WORDS + 1) + w];
        T *dst = dX + (long long)row * C + 4 * q;
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<uint4 *>(dst) = make_uint4(e[0], e[1], e[2], e[3]);
        } else {
            const unsigned sh = 16 * h;
            uint2 v;
            v.x = ((e[0] >> sh) & 0xFFFFu) | (((e[1] >> sh) & 0xFFFFu) << 16);
            v.y = ((e[2] >> sh) & 0xFFFFu) | (((e[3] >> sh) & 0xFFFFu) << 16);
            *reinterpret_cast<uint2 *>(dst) = v;
        }
    }
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

static int mapped_ok(int64_t V, int32_t C, int32_t dtype, int32_t batch) {
    const int pack = dtype == WFS_F32 ? 1 : 2;
    return V > 0 && V % pack == 0 && C >= 4 && C % 4 == 0 && C <= 128 && batch >= 1 && batch <= 65535;
}

extern "C" int wfs_to_dense_mapped(const void *X, const uint32_t *ticket, const int32_t *slot_id, int64_t M,
                                   const int64_t *m_dev, int32_t batch_size, int64_t V, int32_t C, void *Y,
                                   int32_t dtype, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    WFS_REQUIRE(mapped_ok(V, C, dtype, batch_size), WFS_EINVAL, "unsupported shape for the mapped dense(): V=%lld C=%d",
                (long long)V, C);
    WFS_REQUIRE(ticket && slot_id && Y && (M == 0 || X), WFS_EINVAL, "NULL device pointer");
    const dim3 grid((unsigned)wfs_cdiv(V, DM_CELLS), (unsigned)batch_size), block(TB);
    const int words = DM_CELLS / (dtype == WFS_F32 ? 1 : 2);
    const size_t lds = (size_t)C * (words + 1) * sizeof(unsigned);
    const long long *md = (const long long *)m_dev;
    if (dtype == WFS_F32)
        k_to_dense_mapped<float><<<grid, block, lds, stream>>>((const float *)X, ticket, slot_id, M, md, V, C, (float *)Y);
    else        // bf16 and fp16 alike: 2-byte elements, copied as they are
        k_to_dense_mapped<wfs_bf16><<<grid, block, lds, stream>>>((const wfs_bf16 *)X, ticket, slot_id, M, md, V, C,
                                                                  (wfs_bf16 *)Y);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

extern "C" int wfs_to_dense_bwd_mapped(const void *dY, const uint32_t *ticket, const int32_t *slot_id, int64_t M,
                                       const int64_t *m_dev, int32_t batch_size, int64_t V, int32_t C, void *dX,
                                       int32_t dtype, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    WFS_REQUIRE(mapped_ok(V, C, dtype, batch_size), WFS_EINVAL, "unsupported shape for the mapped dense(): V=%lld C=%d",
                (long long)V, C);
    WFS_REQUIRE(ticket && slot_id && dY && (M == 0 || dX), WFS_EINVAL, "NULL device pointer");
    const dim3 grid((unsigned)wfs_cdiv(V, DM_CELLS), (unsigned)batch_size), block(TB);
    const int words = DM_CELLS / (dtype == WFS_F32 ? 1 : 2);
    const size_t lds = (size_t)C * (words + 1) * sizeof(unsigned);
    const long long *md = (const long long *)m_dev;
    if (dtype == WFS_F32)
        k_to_dense_bwd_mapped<float><<<grid, block, lds, stream>>>((const float *)dY, ticket, slot_id, M, md, V, C,
                                                                   (float *)dX);
    else
        k_to_dense_bwd_mapped<wfs_bf16><<<grid, block, lds, stream>>>((const wfs_bf16 *)dY, ticket, slot_id, M, md, V, C,
                                                                      (wfs_bf16 *)dX);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}
