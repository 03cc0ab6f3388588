// head.hip -- the PSD net's classification head: a dense layer with very few outputs.
//
// Reference: SPConvNet.forward flattens ToDense's [B, C, *spatial] to [B, n_linear] and applies the LinearBlock
// (src/models/SPConvNet.py:67-68, src/models/ConvBlocks.py:82-102); with n_type = 2..4 classes the last (often
// the only) nn.Linear is [B, 35840] x [35840, 3].  A GEMM library tiles that for large N and spends ~55 us on a
// 16x16 macro-tile kernel; it is a bandwidth problem (read X once): three small streaming kernels here.
//   forward   Y[b][o] = bias[o] + sum_i X[b][i] W[o][i]          (X fp32 or bf16, W/Y fp32, O <= 8)
//   backward  dX[b][i] = sum_o g[b][o] W[o][i];   dW[o][i] = sum_b g[b][o] X[b][i];   (db is left to the caller)
#include "wfs_common.h"

namespace {
constexpr int TB = 256;
constexpr int MAXO = 8;

template <typename T>
__device__ __forceinline__ void load8(const T *p, float *v) {
    if constexpr (sizeof(T) == 4) {
        float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        uint4 u = *reinterpret_cast<const uint4 *>(p);
        wfs_unpack2<T>(u.x, v[0], v[1]);
        wfs_unpack2<T>(u.y, v[2], v[3]);
        wfs_unpack2<T>(u.z, v[4], v[5]);
        wfs_unpack2<T>(u.w, v[6], v[7]);
    }
}
template <typename T>
__device__ __forceinline__ void store8(T *p, const float *v) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4 *>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        uint4 u;
        u.x = wfs_pack2<T>(v[0], v[1]);
        u.y = wfs_pack2<T>(v[2], v[3]);
        u.z = wfs_pack2<T>(v[4], v[5]);
        u.w = wfs_pack2<T>(v[6], v[7]);
        *reinterpret_cast<uint4 *>(p) = u;
    }
}

// one block per batch row; I % 8 == 0
// NT threads per batch row: 1024 for long rows (four times the loads in flight per row; the row's 17 passes become 5)
template <typename T, int O, int NT>
__global__ void __launch_bounds__(NT) k_head_fwd(const T *__restrict__ X, const float *__restrict__ W,
                                                 const float *__restrict__ bias, float *__restrict__ Y, long long I) {
    __shared__ float red[NT / 64][O];
    const long long b = blockIdx.x;
    float acc[O];
#pragma unroll
    for (int o = 0; o < O; ++o) acc[o] = 0.f;
    const T *x = X + b * I;
    for (long long i = (long long)threadIdx.x * 8; i < I; i += NT * 8) {
        float xv[8];
        load8<T>(x + i, xv);
#pragma unroll
        for (int o = 0; o < O; ++o) {
            float wv[8];
            load8<float>(W + (long long)o * I + i, wv);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o] = fmaf(xv[e], wv[e], acc[o]);
        }
    }
#pragma unroll
    for (int o = 0; o < O; ++o) {
        float v = acc[o];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][o] = v;
    }
    __syncthreads();
    if (threadIdx.x < O) {
        float v = bias ? bias[threadIdx.x] : 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) v += red[w][threadIdx.x];
        Y[b * O + threadIdx.x] = v;
    }
}

// dX: grid = (I / (TB*8) rounded up, B)
template <typename T, int O>
__global__ void __launch_bounds__(TB) k_head_dx(const float *__restrict__ G, const float *__restrict__ W,
                                                T *__restrict__ dX, long long I) {
    const long long b = blockIdx.y;
    const long long i = ((long long)blockIdx.x * TB + threadIdx.x) * 8;
    if (i >= I) return;
    float out[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < O; ++o) {
        float g = G[b * O + o];
        float wv[8];
        load8<float>(W + (long long)o * I + i, wv);
#pragma unroll
        for (int e = 0; e < 8; ++e) out[e] = fmaf(g, wv[e], out[e]);
    }
    store8<T>(dX + b * I + i, out);
}

// dW partials: grid = (I / (TB*8) rounded up, NCHUNK); part[chunk][o][i] = sum over the chunk's batch rows
template <typename T, int O>
__global__ void __launch_bounds__(TB) k_head_dw(const float *__restrict__ G, const T *__restrict__ X,
                                                float *__restrict__ part, long long B, long long I, int rows_per_chunk) {
    const long long i = ((long long)blockIdx.x * TB + threadIdx.x) * 8;
    if (i >= I) return;
    const long long b0 = (long long)blockIdx.y * rows_per_chunk;
    const long long b1 = b0 + rows_per_chunk < B ? b0 + rows_per_chunk : B;
    float acc[O][8];
#pragma unroll
    for (int o = 0; o < O; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[o][e] = 0.f;
#pragma unroll 4
    for (long long b = b0; b < b1; ++b) {
        float xv[8];
        load8<T>(X + b * I + i, xv);
#pragma unroll
        for (int o = 0; o < O; ++o) {
            float g = G[b * O + o];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o][e] = fmaf(g, xv[e], acc[o][e]);
        }
    }
#pragma unroll
    for (int o = 0; o < O; ++o) store8<float>(part + ((long long)blockIdx.y * O + o) * I + i, acc[o]);
}

// dX and the dW partials in ONE launch: blockIdx.y < B serves batch row y of dX, the next NCHUNK values of y the dW
// chunks (same bodies as k_head_dx / k_head_dw); the first dW block also takes the bias gradient dB[o] = sum_b G[b][o].
template <typename T, int O>
__global__ void __launch_bounds__(TB) k_head_bwd(const float *__restrict__ G, const float *__restrict__ W,
                                                 const T *__restrict__ X, T *__restrict__ dX, float *__restrict__ part,
                                                 float *__restrict__ dB, long long B, long long I, int rows_per_chunk,
                                                 int dx_rows) {
    const long long i = ((long long)blockIdx.x * TB + threadIdx.x) * 8;
    if ((int)blockIdx.y < dx_rows) {
        if (i >= I) return;
        const long long b = blockIdx.y;
        float out[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < O; ++o) {
            float g = G[b * O + o];
            float wv[8];
            load8<float>(W + (long long)o * I + i, wv);
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = fmaf(g, wv[e], out[e]);
        }
        store8<T>(dX + b * I + i, out);
        return;
    }
    const int chunk = (int)blockIdx.y - dx_rows;
    if (dB && chunk == 0 && blockIdx.x == 0) {          // all TB threads: strided rows, butterfly, waves in wave order
        __shared__ float sdb[TB / 64][MAXO];
        for (int o = 0; o < O; ++o) {
            float v = 0.f;
            for (long long b = threadIdx.x; b < B; b += TB) v += G[b * O + o];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if ((threadIdx.x & 63) == 0) sdb[threadIdx.x >> 6][o] = v;
        }
        __syncthreads();
        if ((int)threadIdx.x < O) {
            float v = 0.f;
            for (int w = 0; w < TB / 64; ++w) v += sdb[w][threadIdx.x];
            dB[threadIdx.x] = v;
        }
    }
    if (i >= I) return;
    const long long b0 = (long long)chunk * rows_per_chunk;
    const long long b1 = b0 + rows_per_chunk < B ? b0 + rows_per_chunk : B;
    float acc[O][8];
#pragma unroll
    for (int o = 0; o < O; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[o][e] = 0.f;
#pragma unroll 4
    for (long long b = b0; b < b1; ++b) {
        float xv[8];
        load8<T>(X + b * I + i, xv);
#pragma unroll
        for (int o = 0; o < O; ++o) {
            float g = G[b * O + o];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o][e] = fmaf(g, xv[e], acc[o][e]);
        }
    }
#pragma unroll
    for (int o = 0; o < O; ++o) store8<float>(part + ((long long)chunk * O + o) * I + i, acc[o]);
}

// also the bias gradient dB[o] = sum_b G[b][o] (block 0; saves the caller a reduction launch)
__global__ void k_head_dw_reduce(const float *__restrict__ part, int nchunk, long long OI, float *__restrict__ dW,
                                 const float *__restrict__ G, long long B, int O, float *__restrict__ dB) {
    if (dB && blockIdx.x == 0) {                    // all TB threads: strided rows, butterfly, waves in wave order
        __shared__ float sdb[TB / 64][MAXO];
        for (int o = 0; o < O; ++o) {
            float v = 0.f;
            for (long long b = threadIdx.x; b < B; b += TB) v += G[b * O + o];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if ((threadIdx.x & 63) == 0) sdb[threadIdx.x >> 6][o] = v;
        }
        __syncthreads();
        if ((int)threadIdx.x < O) {
            float v = 0.f;
            for (int w = 0; w < TB / 64; ++w) v += sdb[w][threadIdx.x];
            dB[threadIdx.x] = v;
        }
    }
    long long e = (long long)blockIdx.x * TB + threadIdx.x;
    if (e >= OI) return;
    float s = 0.f;
    for (int c = 0; c < nchunk; ++c) s += part[(long long)c * OI + e];
    dW[e] = s;
}

int head_chunks(long long B) {
    long long c = (B + 31) / 32;
    return (int)(c < 1 ? 1 : (c > 16 ? 16 : c));
}
// CrossEntropyLoss(reduction='mean'), forward and the gradient w.r.t. the logits in ONE block: pass 1 gives every
// row's  lse - z[target]  (max-shifted log-sum-exp) and the number of counted rows, a fixed-order tree reduction gives
// the mean, pass 2 writes (softmax - onehot) / count.  Rows whose target equals ignore_index count for nothing.
constexpr int XE_TB = 1024;
__global__ void __launch_bounds__(XE_TB) k_xent_mean(const float *__restrict__ Z, const long long *__restrict__ target,
                                                     long long B, int C, long long ignore_index,
                                                     float *__restrict__ loss, float *__restrict__ dZ) {
    __shared__ float sL[XE_TB];
    __shared__ float sN[XE_TB];
    float l = 0.f, n = 0.f;
    for (long long b = threadIdx.x; b < B; b += XE_TB) {
        const long long t = target[b];
        if (t == ignore_index || t < 0 || t >= C) continue;
        const float *z = Z + b * C;
        float m = z[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(z[c] - m);
        l += (m + logf(se)) - z[t];
        n += 1.f;
    }
    sL[threadIdx.x] = l;
    sN[threadIdx.x] = n;
    __syncthreads();
    for (int w = XE_TB / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            sL[threadIdx.x] += sL[threadIdx.x + w];
            sN[threadIdx.x] += sN[threadIdx.x + w];
        }
        __syncthreads();
    }
    const float cnt = sN[0];
    if (threadIdx.x == 0) *loss = sL[0] / cnt;              // 0/0 = NaN when every row is ignored, as torch
    if (!dZ) return;
    const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
    for (long long b = threadIdx.x; b < B; b += XE_TB) {
        const long long t = target[b];
        const float *z = Z + b * C;
        float *d = dZ + b * C;
        if (t == ignore_index || t < 0 || t >= C) {
            for (int c = 0; c < C; ++c) d[c] = 0.f;
            continue;
        }
        float m = z[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(z[c] - m);
        const float r = inv / se;
        for (int c = 0; c < C; ++c) d[c] = expf(z[c] - m) * r - (c == t ? inv : 0.f);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// ToDense + flatten + Linear without the dense tensor.  The reference densifies the last sparse layer's [M, C] rows
// into [B, C, *spatial] (18 MB of mostly zeros at the PSD shapes), flattens and multiplies with W [O, C V]
// (src/models/SPConvNet.py:65-68).  With distinct output sites that is
//     Y[b][o] = bias[o] + sum_{rows r of event b} sum_c X[r][c] W[o][c V + cell(r)],       V = prod(spatial)
// Rows of one event are contiguous (a conv numbers its outputs in first-seen order over inputs sorted by event, and
// collate_fn sorts the inputs): block b finds its row range by a two-level counting search on the batch column.
//   forward  one block per event; also writes grid[b][cell] = row or -1 (every cell) for the dW pass
//   dX       row-parallel:  dX[r][c] = sum_o g[b][o] W[o][c V + cell]
//   dW       dW[o][c V + cell] = sum_b g[b][o] X[grid[b][cell]][c]: thread = (cell, 8-channel chunk, slice of events),
//            slices combined by a wave butterfly; dB by block 0.  No float atomics anywhere: results are reproducible.
struct HShape {
    int ndim;
    int spatial[4];
};
__device__ __forceinline__ int cell_of(const HShape &sh, const int *row) {
    int lin = 0;
    for (int d = 0; d < sh.ndim; ++d) lin = lin * sh.spatial[d] + row[1 + d];
    return lin;
}
template <typename T, int O>
__global__ void __launch_bounds__(TB) k_sparse_head_fwd(const T *__restrict__ X, const int *__restrict__ idx, long long Mcap,
                                                        const long long *__restrict__ m_dev, HShape sh, int V, int C,
                                                        const float *__restrict__ W, const float *__restrict__ bias,
                                                        float *__restrict__ Y, int *__restrict__ grid) {
    extern __shared__ int sgrid[];                     // [V]
    __shared__ float red[TB / 64][O];
    __shared__ long long srange[2];
    const int b = blockIdx.x;
    const int stride = sh.ndim + 1;
    const long long M = m_dev ? (*m_dev < Mcap ? *m_dev : Mcap) : Mcap;
    // row range [r0, r1) of event b = number of rows with batch < b and < b + 1.  Two-level counting search, all
    // threads: TB equidistant samples bracket both bounds (one memory round trip), the two brackets are then counted
    // exactly (one more) -- a plain binary search would be ~17 dependent loads per block.
    for (int v = threadIdx.x; v < V; v += TB) sgrid[v] = -1;
    const long long step = (M + TB - 1) / TB > 0 ? (M + TB - 1) / TB : 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ int swc[2][TB / 64];
    {
        const long long pos = (long long)threadIdx.x * step;           // first row of this thread's segment
        const int vb = pos < M ? idx[pos * stride] : 0x7FFFFFFF;
        // segments whose FIRST row is below the bound: ballots + a 4-entry LDS sum (LDS atomics on one address
        // would serialise all 256 threads)
        const int c0 = __popcll(__ballot(vb < b)), c1 = __popcll(__ballot(vb < b + 1));
        if (lane == 0) {
            swc[0][wv] = c0;
            swc[1][wv] = c1;
        }
    }
    __syncthreads();
    int n0 = 0, n1 = 0;
#pragma unroll
    for (int w = 0; w < TB / 64; ++w) {
        n0 += swc[0][w];
        n1 += swc[1][w];
    }
    const int seg0 = n0 > 0 ? n0 - 1 : 0, seg1 = n1 > 0 ? n1 - 1 : 0;       // the segments holding the two bounds
    __syncthreads();
    {
        int c0 = 0, c1 = 0;
        for (long long i = threadIdx.x; i < step; i += TB) {
            const long long p0 = (long long)seg0 * step + i, p1 = (long long)seg1 * step + i;
            const int v0 = p0 < M ? idx[p0 * stride] : 0x7FFFFFFF, v1 = p1 < M ? idx[p1 * stride] : 0x7FFFFFFF;
            c0 += v0 < b ? 1 : 0;
            c1 += v1 < b + 1 ? 1 : 0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            c0 += __shfl_xor(c0, off, 64);
            c1 += __shfl_xor(c1, off, 64);
        }
        if (lane == 0) {
            swc[0][wv] = c0;
            swc[1][wv] = c1;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int t0 = 0, t1 = 0;
        for (int w = 0; w < TB / 64; ++w) {
            t0 += swc[0][w];
            t1 += swc[1][w];
        }
        srange[0] = (long long)seg0 * step + t0;
        srange[1] = (long long)seg1 * step + t1;
    }
    __syncthreads();
    const long long r0 = srange[0], r1 = srange[1];
    for (long long r = r0 + threadIdx.x; r < r1; r += TB) sgrid[cell_of(sh, idx + r * stride)] = (int)r;
    __syncthreads();
    if (grid)
        for (int v = threadIdx.x; v < V; v += TB) grid[(long long)b * V + v] = sgrid[v];
    // thread = (row slot, 8-channel chunk): chunks = C / 8
    const int chunks = C >> 3, slots = TB / chunks;
    const int ch = threadIdx.x % chunks, slot = threadIdx.x / chunks;
    float acc[O];
#pragma unroll
    for (int o = 0; o < O; ++o) acc[o] = 0.f;
    const long long CV = (long long)C * V;
    if (slot < slots)
        for (long long r = r0 + slot; r < r1; r += slots) {
            const int cell = cell_of(sh, idx + r * stride);
            float xv[8];
            load8<T>(X + r * C + ch * 8, xv);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float *w = W + (long long)(ch * 8 + i) * V + cell;
#pragma unroll
                for (int o = 0; o < O; ++o) acc[o] = fmaf(xv[i], w[o * CV], acc[o]);
            }
        }
    // fixed-order reduction: butterfly inside the wave, waves in wave order
#pragma unroll
    for (int o = 0; o < O; ++o) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[o] += __shfl_xor(acc[o], off, 64);
    }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int o = 0; o < O; ++o) red[threadIdx.x >> 6][o] = acc[o];
    __syncthreads();
    if (threadIdx.x < O) {
        float v = bias ? bias[threadIdx.x] : 0.f;
        for (int w = 0; w < TB / 64; ++w) v += red[w][threadIdx.x];
        Y[(long long)b * O + threadIdx.x] = v;
    }
}

template <typename T, int O>
__global__ void __launch_bounds__(TB) k_sparse_head_dx(const int *__restrict__ idx, long long Mcap,
                                                       const long long *__restrict__ m_dev, HShape sh, int V, int C,
                                                       const float *__restrict__ W, const float *__restrict__ G,
                                                       T *__restrict__ dX) {
    const long long M = m_dev ? (*m_dev < Mcap ? *m_dev : Mcap) : Mcap;
    const int chunks = C >> 3, stride = sh.ndim + 1;
    const long long CV = (long long)C * V;
    for (long long u = (long long)blockIdx.x * TB + threadIdx.x; u < M * chunks; u += (long long)gridDim.x * TB) {
        const long long r = u / chunks;
        const int ch = (int)(u - r * chunks);
        const int *row = idx + r * stride;
        const int b = row[0], cell = cell_of(sh, row);
        float g[O];
#pragma unroll
        for (int o = 0; o < O; ++o) g[o] = G[(long long)b * O + o];
        float out[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float *w = W + (long long)(ch * 8 + i) * V + cell;
            float a = 0.f;
#pragma unroll
            for (int o = 0; o < O; ++o) a = fmaf(g[o], w[o * CV], a);
            out[i] = a;
        }
        store8<T>(dX + r * C + ch * 8, out);
    }
}

// dW: wave = 2 cells x 32 event slices (lane & 31 = slice, lane >> 5 = cell of the pair), block = 4 waves = 8 cells,
// blockIdx.y = 8-channel chunk.  A thread takes events slice, slice + 32, ... eight at a time (all grid entries, then
// all feature rows, unconditional and clamped: two memory round trips per batch); the 32 slices of a cell are summed by
// a butterfly inside the wave (fixed order), lane 0 of each half stores.
template <typename T, int O>
__global__ void __launch_bounds__(TB) k_sparse_head_dw(const T *__restrict__ X, const int *__restrict__ grid, int B, int V,
                                                       int C, const float *__restrict__ G, float *__restrict__ dW,
                                                       float *__restrict__ dB) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int sl = lane & 31;
    const int cell = (blockIdx.x * 4 + wid) * 2 + (lane >> 5), ch = blockIdx.y;
    const int cellc = cell < V ? cell : V - 1;
    float acc[O][8];
#pragma unroll
    for (int o = 0; o < O; ++o)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[o][i] = 0.f;
    for (int b0 = sl; b0 < B; b0 += 32 * 8) {
        int rr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + u * 32;
            rr[u] = grid[(long long)(b < B ? b : 0) * V + cellc];
            if (b >= B || cell >= V) rr[u] = -1;
        }
        float xv[8][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) load8<T>(X + (long long)(rr[u] >= 0 ? rr[u] : 0) * C + ch * 8, xv[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + u * 32;
#pragma unroll
            for (int o = 0; o < O; ++o) {
                const float g = rr[u] >= 0 ? G[(long long)(b < B ? b : 0) * O + o] : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[o][i] = fmaf(g, xv[u][i], acc[o][i]);
            }
        }
    }
    const long long CV = (long long)C * V;
#pragma unroll
    for (int o = 0; o < O; ++o)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = acc[o][i];
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (sl == 0 && cell < V) dW[o * CV + (long long)(ch * 8 + i) * V + cell] = v;
        }
    if (dB && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < O) {
        float v = 0.f;
        for (int b = 0; b < B; ++b) v += G[(long long)b * O + threadIdx.x];
        dB[threadIdx.x] = v;
    }
}

}  // namespace

extern "C" size_t wfs_head_workspace_bytes(int64_t B, int64_t I, int32_t O) {
    return (size_t)head_chunks(B) * O * I * sizeof(float);
}

#define WFS_HEAD_DISPATCH(O, CALL)                    \
    switch (O) {                                      \
        case 1: { constexpr int OO = 1; CALL; } break; \
        case 2: { constexpr int OO = 2; CALL; } break; \
        case 3: { constexpr int OO = 3; CALL; } break; \
        case 4: { constexpr int OO = 4; CALL; } break; \
        case 5: { constexpr int OO = 5; CALL; } break; \
        case 6: { constexpr int OO = 6; CALL; } break; \
        case 7: { constexpr int OO = 7; CALL; } break; \
        default: { constexpr int OO = 8; CALL; } break; \
    }

extern "C" int wfs_head_fwd(const void *X, int64_t B, int64_t I, const float *W, const float *bias, int32_t O,
                            float *Y, int32_t dtype, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(O >= 1 && O <= MAXO && I % 8 == 0 && I > 0, WFS_EINVAL, "head: need 1 <= O <= 8 and I %% 8 == 0");
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    if (B == 0) return WFS_OK;
    WFS_REQUIRE(X && W && Y, WFS_EINVAL, "NULL device pointer");
    dim3 grid((unsigned)B);
#define WFS_HEAD_FWD(T, NT)                                                                                          \
    WFS_HEAD_DISPATCH(O, (k_head_fwd<T, OO, NT><<<grid, dim3(NT), 0, stream>>>((const T *)X, W, bias, Y, I)))
    if (I >= 8192) {
        if (dtype == WFS_F32) { WFS_HEAD_FWD(float, 1024); } else if (dtype == WFS_BF16) { WFS_HEAD_FWD(wfs_bf16, 1024); } else { WFS_HEAD_FWD(wfs_f16, 1024); }
    } else {
        if (dtype == WFS_F32) { WFS_HEAD_FWD(float, 256); } else if (dtype == WFS_BF16) { WFS_HEAD_FWD(wfs_bf16, 256); } else { WFS_HEAD_FWD(wfs_f16, 256); }
    }
#undef WFS_HEAD_FWD
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

extern "C" int wfs_head_bwd(const void *X, const float *G, int64_t B, int64_t I, const float *W, int32_t O, void *dX,
                            float *dW, float *dB, int32_t dtype, void *workspace, size_t workspace_bytes,
                            wfs_dw_job *defer, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(O >= 1 && O <= MAXO && I % 8 == 0 && I > 0, WFS_EINVAL, "head: need 1 <= O <= 8 and I %% 8 == 0");
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    if (B == 0) {
        if (dW) WFS_HIP_CHECK(hipMemsetAsync(dW, 0, (size_t)O * I * sizeof(float), stream));
        return WFS_OK;
    }
    WFS_REQUIRE(X && G && W, WFS_EINVAL, "NULL device pointer");
    const unsigned gx = (unsigned)wfs_cdiv(I, TB * 8);
    if (defer) *defer = wfs_dw_job{nullptr, 0, 0, 0, 0, 0, 0, nullptr};
    if (dX && dW && B + 16 <= 65535) {
        // one launch for dX and the dW partials (+ dB); the reduction over the partials runs now, or later as one of the
        // caller's deferred jobs (wfs_dw_reduce_jobs)
        const int nchunk = head_chunks(B);
        WFS_REQUIRE(workspace && workspace_bytes >= wfs_head_workspace_bytes(B, I, O), WFS_EWORKSPACE, "workspace too small");
        const int rpc = (int)wfs_cdiv(B, nchunk);
        dim3 grid(gx, (unsigned)(B + nchunk)), block(TB);
        float *part = (float *)workspace;
#define WFS_HEAD_BWD(T)                                                                                              \
    WFS_HEAD_DISPATCH(O, (k_head_bwd<T, OO><<<grid, block, 0, stream>>>(G, W, (const T *)X, (T *)dX, part, dB, B, I, rpc, \
                                                                       (int)B)))
        if (dtype == WFS_F32) { WFS_HEAD_BWD(float); } else if (dtype == WFS_BF16) { WFS_HEAD_BWD(wfs_bf16); } else { WFS_HEAD_BWD(wfs_f16); }
#undef WFS_HEAD_BWD
        WFS_LAUNCH_CHECK();
        const long long OI = (long long)O * I;
        if (defer) {
            *defer = wfs_dw_job{part, nchunk, OI, 1, 1, 1, 0, dW};
            return WFS_OK;
        }
        k_head_dw_reduce<<<dim3((unsigned)wfs_cdiv(OI, TB)), dim3(TB), 0, stream>>>(part, nchunk, OI, dW, G, B, O, nullptr);
        WFS_LAUNCH_CHECK();
        return WFS_OK;
    }
    if (dX) {
        dim3 grid(gx, (unsigned)B), block(TB);
        if (dtype == WFS_F32) {
            WFS_HEAD_DISPATCH(O, (k_head_dx<float, OO><<<grid, block, 0, stream>>>(G, W, (float *)dX, I)));
        } else if (dtype == WFS_BF16) {
            WFS_HEAD_DISPATCH(O, (k_head_dx<wfs_bf16, OO><<<grid, block, 0, stream>>>(G, W, (wfs_bf16 *)dX, I)));
        } else {
            WFS_HEAD_DISPATCH(O, (k_head_dx<wfs_f16, OO><<<grid, block, 0, stream>>>(G, W, (wfs_f16 *)dX, I)));
        }
        WFS_LAUNCH_CHECK();
    }
    if (dW) {
        const int nchunk = head_chunks(B);
        WFS_REQUIRE(workspace && workspace_bytes >= wfs_head_workspace_bytes(B, I, O), WFS_EWORKSPACE, "workspace too small");
        const int rpc = (int)wfs_cdiv(B, nchunk);
        dim3 grid(gx, (unsigned)nchunk), block(TB);
        float *part = (float *)workspace;
        if (dtype == WFS_F32) {
            WFS_HEAD_DISPATCH(O, (k_head_dw<float, OO><<<grid, block, 0, stream>>>(G, (const float *)X, part, B, I, rpc)));
        } else if (dtype == WFS_BF16) {
            WFS_HEAD_DISPATCH(O, (k_head_dw<wfs_bf16, OO><<<grid, block, 0, stream>>>(G, (const wfs_bf16 *)X, part, B, I, rpc)));
        } else {
            WFS_HEAD_DISPATCH(O, (k_head_dw<wfs_f16, OO><<<grid, block, 0, stream>>>(G, (const wfs_f16 *)X, part, B, I, rpc)));
        }
        WFS_LAUNCH_CHECK();
        const long long OI = (long long)O * I;
        k_head_dw_reduce<<<dim3((unsigned)wfs_cdiv(OI, TB)), dim3(TB), 0, stream>>>(part, nchunk, OI, dW, G, B, O, dB);
        WFS_LAUNCH_CHECK();
    }
    return WFS_OK;
}

extern "C" int wfs_xent_mean_fwd_bwd(const float *logits, const int64_t *target, int64_t B, int32_t C,
                                     int64_t ignore_index, float *loss, float *dlogits, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(B >= 1 && C >= 1 && C <= 4096, WFS_EINVAL, "bad logits shape [%lld, %d]", (long long)B, C);
    WFS_REQUIRE(logits && target && loss, WFS_EINVAL, "NULL device pointer");
    k_xent_mean<<<dim3(1), dim3(XE_TB), 0, stream>>>(logits, (const long long *)target, B, C, ignore_index, loss, dlogits);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

static int make_hshape(HShape *sh, int32_t ndim, const int32_t *spatial, long long *V) {
    WFS_REQUIRE(ndim >= 1 && ndim <= 4 && spatial, WFS_EINVAL, "ndim %d not in [1,4]", ndim);
    sh->ndim = ndim;
    *V = 1;
    for (int d = 0; d < 4; ++d) {
        sh->spatial[d] = d < ndim ? spatial[d] : 1;
        *V *= sh->spatial[d];
    }
    return WFS_OK;
}

#define WFS_HEAD_O(CALL)            \
    switch (O) {                    \
        case 1: CALL(1); break;     \
        case 2: CALL(2); break;     \
        case 3: CALL(3); break;     \
        case 4: CALL(4); break;     \
        case 5: CALL(5); break;     \
        case 6: CALL(6); break;     \
        case 7: CALL(7); break;     \
        default: CALL(8); break;    \
    }

extern "C" int wfs_sparse_head_fwd(const void *X, const int32_t *indices, int64_t M, int32_t ndim,
                                   const int32_t *spatial, int32_t batch, int32_t C, const float *W, const float *bias,
                                   int32_t O, float *Y, int32_t *grid, int32_t dtype, const int64_t *m_dev_,
                                   void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const long long *m_dev = (const long long *)m_dev_;
    HShape sh;
    long long V;
    int rc = make_hshape(&sh, ndim, spatial, &V);
    if (rc != WFS_OK) return rc;
    WFS_REQUIRE(O >= 1 && O <= MAXO && C >= 8 && C % 8 == 0 && C / 8 <= TB, WFS_EINVAL, "unsupported head shape C=%d O=%d", C, O);
    WFS_REQUIRE(V >= 1 && V * 4 <= 64 * 1024, WFS_EINVAL, "%lld cells per event do not fit the LDS grid", V);
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    if (batch == 0) return WFS_OK;
    WFS_REQUIRE((M == 0 || (X && indices)) && W && Y, WFS_EINVAL, "NULL device pointer");
    const dim3 grd((unsigned)batch), block(TB);
    const size_t lds = (size_t)V * sizeof(int);
#define WFS_CALL(OO)                                                                                                   \
    if (dtype == WFS_F32)                                                                                              \
        k_sparse_head_fwd<float, OO><<<grd, block, lds, stream>>>((const float *)X, indices, M, m_dev, sh, (int)V, C, W, \
                                                                   bias, Y, grid);                                     \
    else if (dtype == WFS_BF16) \
        k_sparse_head_fwd<wfs_bf16, OO><<<grd, block, lds, stream>>>((const wfs_bf16 *)X, indices, M, m_dev, sh, (int)V, \
                                                                      C, W, bias, Y, grid); \
    else \
        k_sparse_head_fwd<wfs_f16, OO><<<grd, block, lds, stream>>>((const wfs_f16 *)X, indices, M, m_dev, sh, (int)V, \
                                                                      C, W, bias, Y, grid)
    WFS_HEAD_O(WFS_CALL)
#undef WFS_CALL
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

extern "C" int wfs_sparse_head_bwd(const void *X, const int32_t *indices, int64_t M, int32_t ndim,
                                   const int32_t *spatial, int32_t batch, int32_t C, const float *W, int32_t O,
                                   const float *G, void *dX, float *dW, float *dB, const int32_t *grid, int32_t dtype,
                                   const int64_t *m_dev_, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const long long *m_dev = (const long long *)m_dev_;
    HShape sh;
    long long V;
    int rc = make_hshape(&sh, ndim, spatial, &V);
    if (rc != WFS_OK) return rc;
    WFS_REQUIRE(O >= 1 && O <= MAXO && C >= 8 && C % 8 == 0, WFS_EINVAL, "unsupported head shape C=%d O=%d", C, O);
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    WFS_REQUIRE(W && G, WFS_EINVAL, "NULL device pointer");
    if (dX && M > 0) {
        WFS_REQUIRE(indices, WFS_EINVAL, "NULL indices");
        long long blocks = wfs_cdiv(M * (C / 8), TB);
        if (blocks > 4096) blocks = 4096;
        const dim3 grd((unsigned)blocks), block(TB);
#define WFS_CALL(OO)                                                                                                  \
    if (dtype == WFS_F32)                                                                                             \
        k_sparse_head_dx<float, OO><<<grd, block, 0, stream>>>(indices, M, m_dev, sh, (int)V, C, W, G, (float *)dX);   \
    else if (dtype == WFS_BF16)                                                                                       \
        k_sparse_head_dx<wfs_bf16, OO><<<grd, block, 0, stream>>>(indices, M, m_dev, sh, (int)V, C, W, G, (wfs_bf16 *)dX); \
    else                                                                                                              \
        k_sparse_head_dx<wfs_f16, OO><<<grd, block, 0, stream>>>(indices, M, m_dev, sh, (int)V, C, W, G, (wfs_f16 *)dX)
        WFS_HEAD_O(WFS_CALL)
#undef WFS_CALL
        WFS_LAUNCH_CHECK();
    }
    if (dW) {
        WFS_REQUIRE(grid && (X || M == 0), WFS_EINVAL, "dW needs the forward's grid and the features");
        const dim3 grd((unsigned)wfs_cdiv(V, 8), (unsigned)(C / 8)), block(TB);
#define WFS_CALL(OO)                                                                                                  \
    if (dtype == WFS_F32)                                                                                             \
        k_sparse_head_dw<float, OO><<<grd, block, 0, stream>>>((const float *)X, grid, batch, (int)V, C, G, dW, dB);   \
    else if (dtype == WFS_BF16)                                                                                       \
        k_sparse_head_dw<wfs_bf16, OO><<<grd, block, 0, stream>>>((const wfs_bf16 *)X, grid, batch, (int)V, C, G, dW, dB); \
    else                                                                                                              \
        k_sparse_head_dw<wfs_f16, OO><<<grd, block, 0, stream>>>((const wfs_f16 *)X, grid, batch, (int)V, C, G, dW, dB)
        WFS_HEAD_O(WFS_CALL)
#undef WFS_CALL
        WFS_LAUNCH_CHECK();
    }
    return WFS_OK;
}
