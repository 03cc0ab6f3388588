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

// ---- rows of any length (scalar loads): the short second layer of a two-layer head (the hybrid net's Linear(269, 3),
// reference src/models/SPConvNet.py:40-52) -- a few hundred inputs, nothing a library GEMM or a streaming kernel is for
template <typename T, int O>
__global__ void __launch_bounds__(64) k_head_fwd_any(const T *__restrict__ X, const float *__restrict__ W,
                                                     const float *__restrict__ bias, float *__restrict__ Y, long long I) {
    const long long b = blockIdx.x;
    const int lane = threadIdx.x;
    float acc[O];
#pragma unroll
    for (int o = 0; o < O; ++o) acc[o] = 0.f;
    for (long long i = lane; i < I; i += 64) {
        const float x = wfs_ld(X + b * I + i);
#pragma unroll
        for (int o = 0; o < O; ++o) acc[o] = fmaf(x, W[o * I + i], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < O; ++o) {
        float v = acc[o];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);       // fixed tree
        if (lane == 0) Y[b * O + o] = v + (bias ? bias[o] : 0.f);
    }
}

// dX[b][i] = sum_o G[b][o] W[o][i]: one thread per element
template <typename T, int O>
__global__ void __launch_bounds__(TB) k_head_dx_any(const float *__restrict__ G, const float *__restrict__ W,
                                                    T *__restrict__ dX, long long B, long long I) {
    const long long e = (long long)blockIdx.x * TB + threadIdx.x;
    if (e >= B * I) return;
    const long long b = e / I, i = e % I;
    float v = 0.f;
#pragma unroll
    for (int o = 0; o < O; ++o) v = fmaf(G[b * O + o], W[o * I + i], v);
    wfs_st(dX + e, v);
}

// dW[o][i] = sum_b G[b][o] X[b][i]; the column past the last input takes dB[o] = sum_b G[b][o].  Block = 16 columns x
// 16 batch slots (slot s adds rows s, s + 16, ... in order; the slots are added in slot order: fixed, no atomics).
template <typename T, int O>
__global__ void __launch_bounds__(TB) k_head_dw_any(const float *__restrict__ G, const T *__restrict__ X,
                                                    float *__restrict__ dW, float *__restrict__ dB, long long B,
                                                    long long I) {
    __shared__ float red[16][16][O];
    const int col = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const long long i = (long long)blockIdx.x * 16 + col;
    float acc[O];
#pragma unroll
    for (int o = 0; o < O; ++o) acc[o] = 0.f;
    if (i <= I) {
#pragma unroll 4
        for (long long b = slot; b < B; b += 16) {
            const float x = i < I ? wfs_ld(X + b * I + i) : 1.f;
#pragma unroll
            for (int o = 0; o < O; ++o) acc[o] = fmaf(G[b * O + o], x, acc[o]);
        }
    }
#pragma unroll
    for (int o = 0; o < O; ++o) red[slot][col][o] = acc[o];
    __syncthreads();
    if (slot == 0 && i <= I) {
#pragma unroll
        for (int o = 0; o < O; ++o) {
            float v = red[0][col][o];
            for (int q = 1; q < 16; ++q) v += red[q][col][o];
            if (i < I) dW[o * I + i] = v;
            else if (dB) dB[o] = v;
        }
    }
}

// rows the vector kernels do not take (I % 8 != 0) or that are too short to be worth their chunked reduction
inline bool head_any(long long I) { return I % 8 != 0 || I < 1024; }

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
    WFS_REQUIRE(O >= 1 && O <= MAXO && I > 0, WFS_EINVAL, "head: need 1 <= O <= 8");
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    if (B == 0) return WFS_OK;
    WFS_REQUIRE(X && W && Y, WFS_EINVAL, "NULL device pointer");
    dim3 grid((unsigned)B);
    if (head_any(I)) {
#define WFS_HEAD_FWD_ANY(T) WFS_HEAD_DISPATCH(O, (k_head_fwd_any<T, OO><<<grid, dim3(64), 0, stream>>>((const T *)X, W, bias, Y, I)))
        if (dtype == WFS_F32) { WFS_HEAD_FWD_ANY(float); } else if (dtype == WFS_BF16) { WFS_HEAD_FWD_ANY(wfs_bf16); } else { WFS_HEAD_FWD_ANY(wfs_f16); }
#undef WFS_HEAD_FWD_ANY
        WFS_LAUNCH_CHECK();
        return WFS_OK;
    }
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
    WFS_REQUIRE(O >= 1 && O <= MAXO && I > 0, WFS_EINVAL, "head: need 1 <= O <= 8");
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    if (B == 0) {
        if (dW) WFS_HIP_CHECK(hipMemsetAsync(dW, 0, (size_t)O * I * sizeof(float), stream));
        return WFS_OK;
    }
    WFS_REQUIRE(X && G && W, WFS_EINVAL, "NULL device pointer");
    if (head_any(I)) {
        if (defer) *defer = wfs_dw_job{nullptr, 0, 0, 0, 0, 0, 0, nullptr};
        const dim3 gdx((unsigned)wfs_cdiv(B * I, TB)), gdw((unsigned)wfs_cdiv(I + 1, 16)), block(TB);
#define WFS_HEAD_BWD_ANY(T)                                                                                          \
    do {                                                                                                             \
        if (dX) WFS_HEAD_DISPATCH(O, (k_head_dx_any<T, OO><<<gdx, block, 0, stream>>>(G, W, (T *)dX, B, I)));        \
        if (dW) WFS_HEAD_DISPATCH(O, (k_head_dw_any<T, OO><<<gdw, block, 0, stream>>>(G, (const T *)X, dW, dB, B, I))); \
    } while (0)
        if (dtype == WFS_F32) { WFS_HEAD_BWD_ANY(float); } else if (dtype == WFS_BF16) { WFS_HEAD_BWD_ANY(wfs_bf16); } else { WFS_HEAD_BWD_ANY(wfs_f16); }
#undef WFS_HEAD_BWD_ANY
        WFS_LAUNCH_CHECK();
        return WFS_OK;
    }
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
