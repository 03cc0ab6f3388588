// tcn.hip -- the hybrid net's waveform front end in one launch per direction.
//
// Reference: SPConvNet puts TemporalConvNet(1, [1] * n_dil, kernel_size, dropout) in front of the sparse stack
// (src/models/SPConvNet.py:56-61,83-92); the TCN (src/models/ConvBlocks.py:114-173, the locuslab design) is, per
// level i with dilation d = 2^i, two weight-normed Conv1d(1 -> 1, k, dilation d, padding (k-1) d) each chomped on the
// right (= causal), ReLU after each, and a residual:  x_{i+1} = relu(relu(conv2(relu(conv1(x_i)))) + x_i).
// With ONE channel a level is two k-tap causal FIR filters per row, and the rows [N, L = 2 T samples] are independent:
// torch runs ~8 launches per level, each a full pass over [N, L] in HBM; here a block keeps its row in LDS through all
// levels -- one read and one write of the row per direction (HBM-bound: 2 N L s bytes forward, 3 N L s backward).
//
//   forward   h1[t] = relu(b1 + sum_j w1[j] x[t - (k-1-j) d]),  h2 likewise from h1,  x' = relu(h2 + x)
//   backward  recomputes the activations of all levels into LDS ((3 levels + 1) rows), then walks the levels down;
//             per-row partial sums of dW / dB are reduced by the caller in a fixed order.
// Weight norm (w = g v / |v|) stays with the caller: the kernels take the effective taps (device memory).
// Dropout (nn.Dropout(p) after each of the two ReLUs of a level, ConvBlocks.py:125-134, active in training) is applied
// in place: the keep/drop decision of element (row, level, conv, t) is a counter-based hash of a 64-bit seed that the
// caller draws from torch's generator into device memory -- nothing is stored, the backward's recompute sees the same
// masks.  (The masks are this library's own stream of random numbers, not torch's: same distribution, other bits.)
#include "wfs_common.h"

namespace {

constexpr int TB = 256;
constexpr int TBW = 1024;         // the backward kernel's block (one block per CU: make it wide)
constexpr int MAXLV = 8, MAXK = 8;

// the effective taps [levels][2][k] and biases [levels][2] are autograd tensors in device memory: every block
// copies them into LDS first (a few dozen floats, no host round trip, capturable in a HIP graph)
struct Taps {
    float w[MAXLV * 2 * MAXK];
    float b[MAXLV * 2];
};
template <int K>
__device__ __forceinline__ void load_taps(Taps *tp, const float *W, const float *B, int levels) {
    for (int i = threadIdx.x; i < levels * 2 * K; i += TB) tp->w[i] = W[i];
    for (int i = threadIdx.x; i < levels * 2; i += TB) tp->b[i] = B[i];
}

// dropout multiplier of element t of conv `ci` (= 2 * level + {0, 1}) of `row`: 0 with probability p, else 1 / (1 - p).
// splitmix64 finaliser over a counter that is unique per element (t < 2^12, ci < 2^4).
struct Drop {
    unsigned long long seed;
    unsigned threshold;      // drop when the hash's high 32 bits are below p * 2^32
    float scale;             // 1 / (1 - p); 1 when dropout is off
    bool on;
};
__device__ __forceinline__ Drop make_drop(float p, const long long *seed_dev) {
    Drop d;
    d.on = p > 0.f && seed_dev != nullptr;
    d.seed = d.on ? (unsigned long long)*seed_dev : 0ull;
    double th = (double)p * 4294967296.0;
    d.threshold = th >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)th;
    d.scale = d.on ? 1.f / (1.f - p) : 1.f;
    return d;
}
__device__ __forceinline__ float drop_mult(const Drop &d, long long row, int ci, int t) {
    if (!d.on) return 1.f;
    unsigned long long z = d.seed + (((unsigned long long)row << 16) | ((unsigned long long)ci << 12) | (unsigned)t) *
                                        0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 32) < d.threshold ? 0.f : d.scale;
}

template <typename T>
__device__ __forceinline__ float ldv(const T *p) {
    return wfs_ld(p);
}

// out[t] = bias + sum_j w[j] in[t - (K-1-j) d]   (zero to the left of the row)
template <int K>
__device__ __forceinline__ float fir(const float *in, int t, const float *w, float bias, int d) {
    float a = bias;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        int s = t - (K - 1 - j) * d;
        a = fmaf(w[j], s >= 0 ? in[s] : 0.f, a);
    }
    return a;
}

template <typename T, int K>
__global__ void __launch_bounds__(TB) k_tcn_fwd(const T *__restrict__ X, long long N, int L, const float *__restrict__ Wd,
                                                const float *__restrict__ Bd, int levels, T *__restrict__ Y,
                                                float drop_p, const long long *__restrict__ seed_dev) {
    extern __shared__ float lds[];
    __shared__ Taps tp;
    float *A = lds, *B = lds + L;
    const long long row = blockIdx.x;
    const Drop dr = make_drop(drop_p, seed_dev);
    const T *x = X + row * L;
    load_taps<K>(&tp, Wd, Bd, levels);
    for (int t = threadIdx.x; t < L; t += TB) A[t] = ldv(x + t);
    __syncthreads();
    for (int lv = 0; lv < levels; ++lv) {
        const int d = 1 << lv;
        for (int t = threadIdx.x; t < L; t += TB) {
            float v = fir<K>(A, t, tp.w + (lv * 2 + 0) * K, tp.b[lv * 2 + 0], d);
            B[t] = (v > 0.f ? v : 0.f) * drop_mult(dr, row, lv * 2 + 0, t);
        }
        __syncthreads();
        // x' overwrites A in place: A[t] is read only at index t by the thread that rewrites it
        for (int t = threadIdx.x; t < L; t += TB) {
            float v = fir<K>(B, t, tp.w + (lv * 2 + 1) * K, tp.b[lv * 2 + 1], d);
            v = (v > 0.f ? v : 0.f) * drop_mult(dr, row, lv * 2 + 1, t) + A[t];
            A[t] = v > 0.f ? v : 0.f;
        }
        __syncthreads();
    }
    T *y = Y + row * L;
    for (int t = threadIdx.x; t < L; t += TB) wfs_st(y + t, A[t]);
}

// LDS: Xs[levels + 1][L], H1[levels][L], H2[levels][L], G[L], G2[L], G3[L] -- 13 rows = 106 KB at L = 2048, 3 levels: one
// block per CU, so the block is 16 waves (TBW): a row pass is 2 steps per thread instead of 8 (112 -> ? us at 776 rows)
template <typename T, int K>
__global__ void __launch_bounds__(TBW) k_tcn_bwd(const T *__restrict__ X, const T *__restrict__ dY, long long N, int L,
                                                const float *__restrict__ Wd, const float *__restrict__ Bd, int levels,
                                                T *__restrict__ dX, float *__restrict__ partial, float drop_p,
                                                const long long *__restrict__ seed_dev) {
    constexpr int k = K;
    extern __shared__ float lds[];
    __shared__ Taps tp;
    float *Xs = lds;
    float *H1 = Xs + (size_t)(levels + 1) * L;
    float *H2 = H1 + (size_t)levels * L;
    float *G = H2 + (size_t)levels * L;
    float *G2 = G + L;
    float *G3 = G2 + L;
    __shared__ float sred[TBW];
    const long long row = blockIdx.x;
    const int nthr = blockDim.x;                 // 256 or TBW (the host picks by the LDS a row needs)
    const T *x = X + row * L;
    const Drop dr = make_drop(drop_p, seed_dev);
    load_taps<K>(&tp, Wd, Bd, levels);
    for (int t = threadIdx.x; t < L; t += nthr) {
        Xs[t] = ldv(x + t);
        G[t] = ldv(dY + row * L + t);
    }
    __syncthreads();
    // ---- forward recompute, everything kept
    for (int lv = 0; lv < levels; ++lv) {
        const int d = 1 << lv;
        const float *A = Xs + (size_t)lv * L;
        float *h1 = H1 + (size_t)lv * L, *h2 = H2 + (size_t)lv * L, *An = Xs + (size_t)(lv + 1) * L;
        for (int t = threadIdx.x; t < L; t += nthr) {
            float v = fir<K>(A, t, tp.w + (lv * 2 + 0) * K, tp.b[lv * 2 + 0], d);
            h1[t] = (v > 0.f ? v : 0.f) * drop_mult(dr, row, lv * 2 + 0, t);     // the value conv2 sees
        }
        __syncthreads();
        for (int t = threadIdx.x; t < L; t += nthr) {
            float v = fir<K>(h1, t, tp.w + (lv * 2 + 1) * K, tp.b[lv * 2 + 1], d);
            v = (v > 0.f ? v : 0.f) * drop_mult(dr, row, lv * 2 + 1, t);
            h2[t] = v;
            float o = v + A[t];
            An[t] = o > 0.f ? o : 0.f;
        }
        __syncthreads();
    }
    // ---- backward walk; acc[(lv*2 + c) * (k+1) + j]: dW taps j < k, dB at j == k (per thread, then block-reduced)
    const int per = 2 * (k + 1);
    for (int lv = levels - 1; lv >= 0; --lv) {
        const int d = 1 << lv;
        const float *A = Xs + (size_t)lv * L, *An = Xs + (size_t)(lv + 1) * L;
        const float *h1 = H1 + (size_t)lv * L, *h2 = H2 + (size_t)lv * L;
        float a2[K + 1], a1[K + 1];
#pragma unroll
        for (int j = 0; j <= k; ++j) a2[j] = a1[j] = 0.f;
        // G  <- g_out = G * [x_{lv+1} > 0];   G2 <- g_h2 = g_out * [h2 > 0]
        for (int t = threadIdx.x; t < L; t += nthr) {
            float go = An[t] > 0.f ? G[t] : 0.f;
            float gh = h2[t] > 0.f ? go * dr.scale : 0.f;     // h2 = relu(.) * mult > 0 <=> positive AND kept
            G[t] = go;
            G2[t] = gh;
            a2[k] += gh;
#pragma unroll
            for (int j = 0; j < k; ++j) {
                int s = t - (k - 1 - j) * d;
                a2[j] = fmaf(gh, s >= 0 ? h1[s] : 0.f, a2[j]);
            }
        }
        __syncthreads();
        // g_h1[s] = [h1 > 0] * sum_j w2[j] g_h2[s + (k-1-j) d]  -> G3;  its dW1 / dB1 sums in the same pass
        for (int s = threadIdx.x; s < L; s += nthr) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < k; ++j) {
                int t = s + (k - 1 - j) * d;
                v = fmaf(tp.w[(lv * 2 + 1) * K + j], t < L ? G2[t] : 0.f, v);
            }
            v = h1[s] > 0.f ? v * dr.scale : 0.f;
            G3[s] = v;
            a1[k] += v;
#pragma unroll
            for (int j = 0; j < k; ++j) {
                int u = s - (k - 1 - j) * d;
                a1[j] = fmaf(v, u >= 0 ? A[u] : 0.f, a1[j]);
            }
        }
        __syncthreads();
        // dX_lv[s] = g_out[s] (residual) + sum_j w1[j] g_h1[s + (k-1-j) d]   (G[s] is read only at index s)
        for (int s = threadIdx.x; s < L; s += nthr) {
            float v = G[s];
#pragma unroll
            for (int j = 0; j < k; ++j) {
                int t = s + (k - 1 - j) * d;
                v = fmaf(tp.w[(lv * 2 + 0) * K + j], t < L ? G3[t] : 0.f, v);
            }
            G[s] = v;
        }
        // block reduction of this level's 2 (k + 1) sums in a fixed order: butterfly inside each wave, then the
        // waves' sums added in wave order -- one barrier per level
        {
            constexpr int P = 2 * (K + 1);
            float vals[P];
#pragma unroll
            for (int j = 0; j <= k; ++j) {
                vals[j] = a1[j];
                vals[(K + 1) + j] = a2[j];
            }
#pragma unroll
            for (int i = 0; i < P; ++i) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) vals[i] += __shfl_xor(vals[i], o, 64);
            }
            __syncthreads();                        // sred is free (previous level's readers are done)
            if ((threadIdx.x & 63) == 0) {
#pragma unroll
                for (int i = 0; i < P; ++i) sred[(threadIdx.x >> 6) * P + i] = vals[i];
            }
            __syncthreads();
            if (threadIdx.x < P) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < nthr / 64; ++w) v += sred[w * P + threadIdx.x];
                partial[row * ((long long)levels * per) + lv * per + threadIdx.x] = v;
            }
        }
        __syncthreads();
    }
    T *dx = dX + row * L;
    for (int t = threadIdx.x; t < L; t += nthr) wfs_st(dx + t, G[t]);
}

}  // namespace

extern "C" size_t wfs_tcn_lds_bytes(int32_t L, int32_t levels, int32_t backward) {
    return (size_t)L * sizeof(float) * (backward ? (3 * levels + 4) : 2);
}

#define WFS_TCN_DISPATCH_K(KERNEL, T, ...)                                  \
    switch (k) {                                                            \
        case 1: KERNEL<T, 1><<<grid, block, lds, stream>>>(__VA_ARGS__); break; \
        case 2: KERNEL<T, 2><<<grid, block, lds, stream>>>(__VA_ARGS__); break; \
        case 3: KERNEL<T, 3><<<grid, block, lds, stream>>>(__VA_ARGS__); break; \
        case 4: KERNEL<T, 4><<<grid, block, lds, stream>>>(__VA_ARGS__); break; \
        case 5: KERNEL<T, 5><<<grid, block, lds, stream>>>(__VA_ARGS__); break; \
        case 6: KERNEL<T, 6><<<grid, block, lds, stream>>>(__VA_ARGS__); break; \
        case 7: KERNEL<T, 7><<<grid, block, lds, stream>>>(__VA_ARGS__); break; \
        default: KERNEL<T, 8><<<grid, block, lds, stream>>>(__VA_ARGS__); break; \
    }

extern "C" int wfs_tcn_fwd(const void *X, int64_t N, int32_t L, const float *taps, const float *bias, int32_t levels,
                           int32_t k, void *Y, int32_t dtype, float dropout_p, const int64_t *seed_dev_,
                           void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(levels >= 1 && levels <= MAXLV && k >= 1 && k <= MAXK, WFS_EINVAL, "unsupported TCN shape: %d levels, k = %d",
                levels, k);
    WFS_REQUIRE(L >= 1 && L <= 16 * TB, WFS_EINVAL, "row length %d not in [1, %d]", L, 16 * TB);
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    WFS_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f && (dropout_p == 0.f || seed_dev_), WFS_EINVAL,
                "dropout %g needs 0 <= p < 1 and a seed", (double)dropout_p);
    const long long *seed_dev = (const long long *)seed_dev_;
    if (N == 0) return WFS_OK;
    WFS_REQUIRE(X && Y && taps && bias, WFS_EINVAL, "NULL device pointer");
    const size_t lds = wfs_tcn_lds_bytes(L, levels, 0);
    const dim3 grid((unsigned)N), block(TB);
    if (dtype == WFS_F32) {
        WFS_TCN_DISPATCH_K(k_tcn_fwd, float, (const float *)X, N, L, taps, bias, levels, (float *)Y, dropout_p, seed_dev)
    } else if (dtype == WFS_BF16) {
        WFS_TCN_DISPATCH_K(k_tcn_fwd, wfs_bf16, (const wfs_bf16 *)X, N, L, taps, bias, levels, (wfs_bf16 *)Y, dropout_p, seed_dev)
    } else {
        WFS_TCN_DISPATCH_K(k_tcn_fwd, wfs_f16, (const wfs_f16 *)X, N, L, taps, bias, levels, (wfs_f16 *)Y, dropout_p, seed_dev)
    }
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

template <typename T, int K>
static int tcn_bwd_attr() {
    static bool done = false;
    if (!done) {
        WFS_HIP_CHECK(hipFuncSetAttribute((const void *)k_tcn_bwd<T, K>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        done = true;
    }
    return WFS_OK;
}
template <typename T>
static int tcn_bwd_attr_k(int k) {
    switch (k) {
        case 1: return tcn_bwd_attr<T, 1>();
        case 2: return tcn_bwd_attr<T, 2>();
        case 3: return tcn_bwd_attr<T, 3>();
        case 4: return tcn_bwd_attr<T, 4>();
        case 5: return tcn_bwd_attr<T, 5>();
        case 6: return tcn_bwd_attr<T, 6>();
        case 7: return tcn_bwd_attr<T, 7>();
        default: return tcn_bwd_attr<T, 8>();
    }
}

extern "C" int wfs_tcn_bwd(const void *X, const void *dY, int64_t N, int32_t L, const float *taps, const float *bias,
                           int32_t levels, int32_t k, void *dX, float *partial, int32_t dtype, float dropout_p,
                           const int64_t *seed_dev_, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(levels >= 1 && levels <= MAXLV && k >= 1 && k <= MAXK, WFS_EINVAL, "unsupported TCN shape: %d levels, k = %d",
                levels, k);
    WFS_REQUIRE(L >= 1 && L <= 16 * TB, WFS_EINVAL, "row length %d not in [1, %d]", L, 16 * TB);
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    WFS_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f && (dropout_p == 0.f || seed_dev_), WFS_EINVAL,
                "dropout %g needs 0 <= p < 1 and a seed", (double)dropout_p);
    const long long *seed_dev = (const long long *)seed_dev_;
    const size_t lds = wfs_tcn_lds_bytes(L, levels, 1);
    WFS_REQUIRE(lds <= 150 * 1024, WFS_EINVAL, "row of %d samples x %d levels needs %zu B of LDS", L, levels, lds);
    if (N == 0) return WFS_OK;
    WFS_REQUIRE(X && dY && dX && partial && taps && bias, WFS_EINVAL, "NULL device pointer");
    // rows that leave room for one or two blocks per CU get a 16-wave block, short rows keep 4 waves and more blocks
    const dim3 grid((unsigned)N), block(lds > 48 * 1024 ? TBW : TB);
    if (dtype == WFS_F32) {
        int rc = tcn_bwd_attr_k<float>(k);
        if (rc != WFS_OK) return rc;
        WFS_TCN_DISPATCH_K(k_tcn_bwd, float, (const float *)X, (const float *)dY, N, L, taps, bias, levels, (float *)dX, partial,
                           dropout_p, seed_dev)
    } else if (dtype == WFS_BF16) {
        int rc = tcn_bwd_attr_k<wfs_bf16>(k);
        if (rc != WFS_OK) return rc;
        WFS_TCN_DISPATCH_K(k_tcn_bwd, wfs_bf16, (const wfs_bf16 *)X, (const wfs_bf16 *)dY, N, L, taps, bias, levels,
                           (wfs_bf16 *)dX, partial, dropout_p, seed_dev)
    } else {
        int rc = tcn_bwd_attr_k<wfs_f16>(k);
        if (rc != WFS_OK) return rc;
        WFS_TCN_DISPATCH_K(k_tcn_bwd, wfs_f16, (const wfs_f16 *)X, (const wfs_f16 *)dY, N, L, taps, bias, levels,
                           (wfs_f16 *)dX, partial, dropout_p, seed_dev)
    }
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

// ------------------------------------------------------------------------------------------ weight norm of the taps
// The reference's TemporalBlock wraps its convolutions in torch.nn.utils.weight_norm (src/models/ConvBlocks.py:118-131):
// w = g v / |v| per output channel.  For the single-channel front end that is, per convolution, k taps from k + 1
// parameters -- torch spends ~16 launches per step on it (forward) and more on its backward.  Here: ONE launch gathers
// all convolutions' (v, g, b) through a pointer table into the [n_conv][k] taps / [n_conv] biases the fused kernels take,
// and ONE launch turns the per-row partial sums of the backward into dv, dg, db, written straight to where the caller
// wants them (the parameters' gradient slots).
struct TcnParamPtrs {           // one convolution: device addresses (0 = absent)
    const float *v, *g, *b;
    float *dv, *dg, *db;
};

__global__ void __launch_bounds__(64) k_tcn_taps(const TcnParamPtrs *__restrict__ pp, int n_conv, int k,
                                                 float *__restrict__ taps, float *__restrict__ bias) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n_conv) return;
    const TcnParamPtrs p = pp[c];
    float n2 = 0.f;
    for (int j = 0; j < k; ++j) n2 = fmaf(p.v[j], p.v[j], n2);
    const float scale = p.g[0] / sqrtf(n2);
    for (int j = 0; j < k; ++j) taps[c * k + j] = p.v[j] * scale;
    bias[c] = p.b ? p.b[0] : 0.f;
}

// partial [N][n_conv][k + 1] (d taps, d bias per row, from k_tcn_bwd): block c sums its convolution's columns over the
// rows (256 threads, interleaved rows, LDS tree: a fixed order), then thread 0 applies the weight-norm backward:
//   dg = (dw . v) / |v|,   dv = g / |v| * (dw - v (dw . v) / |v|^2),   db = sum of the bias column
__global__ void __launch_bounds__(256) k_tcn_taps_bwd(const TcnParamPtrs *__restrict__ pp, int n_conv, int k,
                                                      const float *__restrict__ partial, long long N) {
    __shared__ float sR[256][9];
    const int c = blockIdx.x;
    float acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = 0.f;
    const long long stride = (long long)n_conv * (k + 1);
    for (long long n = threadIdx.x; n < N; n += 256) {
        const float *q = partial + n * stride + (long long)c * (k + 1);
#pragma unroll
        for (int j = 0; j < 9; ++j)
            if (j <= k) acc[j] += q[j];
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) sR[threadIdx.x][j] = acc[j];
    __syncthreads();
    for (int half = 128; half >= 1; half >>= 1) {
        if (threadIdx.x < half)
#pragma unroll
            for (int j = 0; j < 9; ++j) sR[threadIdx.x][j] += sR[threadIdx.x + half][j];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const TcnParamPtrs p = pp[c];
    float n2 = 0.f, dot = 0.f;
    for (int j = 0; j < k; ++j) {
        n2 = fmaf(p.v[j], p.v[j], n2);
        dot = fmaf(sR[0][j], p.v[j], dot);
    }
    const float inv = 1.f / sqrtf(n2), g = p.g[0];
    if (p.dg) p.dg[0] = dot * inv;
    if (p.dv)
        for (int j = 0; j < k; ++j) p.dv[j] = g * inv * (sR[0][j] - p.v[j] * dot * inv * inv);
    if (p.db) p.db[0] = sR[0][k];
}

extern "C" int wfs_tcn_taps_fwd(const void *param_ptrs, int32_t n_conv, int32_t k, float *taps, float *bias, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(n_conv >= 1 && n_conv <= 16 && k >= 1 && k <= 8, WFS_EINVAL, "1 .. 16 convolutions of 1 .. 8 taps (%d, %d)", n_conv, k);
    WFS_REQUIRE(param_ptrs && taps && bias, WFS_EINVAL, "NULL device pointer");
    k_tcn_taps<<<dim3(1), dim3(64), 0, stream>>>((const TcnParamPtrs *)param_ptrs, n_conv, k, taps, bias);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

extern "C" int wfs_tcn_taps_bwd(const void *param_ptrs, int32_t n_conv, int32_t k, const float *partial, int64_t N,
                                void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(n_conv >= 1 && n_conv <= 16 && k >= 1 && k <= 8, WFS_EINVAL, "1 .. 16 convolutions of 1 .. 8 taps (%d, %d)", n_conv, k);
    WFS_REQUIRE(param_ptrs && (partial || N == 0), WFS_EINVAL, "NULL device pointer");
    k_tcn_taps_bwd<<<dim3((unsigned)n_conv), dim3(256), 0, stream>>>((const TcnParamPtrs *)param_ptrs, n_conv, k, partial, N);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}
