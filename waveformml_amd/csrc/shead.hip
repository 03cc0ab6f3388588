// shead.hip -- the classification head straight off the sparse rows (round 4).
//
// Replaces, for the reference's SPConvNet tail `spconv.ToDense -> view(-1, n_linear) -> nn.Linear(n_linear, n_type)`
// (src/models/SPConvNet.py:65-68 with the one-layer LinearBlock of src/models/ConvBlocks.py:82-102), the dense detour
// dense() -> Linear -> Linear backward -> dense() backward: at the PSD batch that detour moves an 18-MB tensor that is
// 90 % zeros four times (38.7 us per step, profiles/r03_hipgraph_bf16_step_summary.txt) for 3 MB of rows.
//
//     logits[b][o] = bias[o] + sum over the rows i of event b, channels c:  X[i][c] * W[o][c * V + cell(i)]
//
// nn.Linear's weight is channels-FIRST over the dense grid: the 32 channels of one cell lie V floats apart, so a kernel
// that walks the rows reads 96 scattered floats per row (round 2's attempt: bound by L2 requests).  Here the work is
// CELL-major instead: lane = cell, so W[o][c * V + cell .. cell + 63] is one coalesced read in the layout the parameter
// has, a thread keeps its cell's O x 8 weights in registers, and walks the EVENTS of its slice through the cell -> row
// map the last strided layer's rulebook build left behind (wfs_event_rulebook_conv cell_row / wfs_rulebook_cell_map):
// block = (64 cells x C / 8 channel groups) x (slice of SH_EVENTS events).
//   forward   per (event, output) a sum over the block's cells -> partial[tile][event][output]; k_shead_sum adds the
//             tiles in a fixed order (+ bias)
//   backward  ONE launch: dX[row][c] = sum_o g[b][o] W[o][c V + cell] (every valid row has exactly one cell: written
//             once), dW partials per slice in the parameter's own layout (coalesced over cells; the slices are summed by
//             the step's deferred slab reduction), db
// No atomics, fixed summation orders: run-to-run reproducible.
#include <stdlib.h>

#include "wfs_common.h"

namespace {

constexpr int SH_CELLS = 64;          // cells per block = lanes of a wave
constexpr int SH_EVENTS = 16;         // events per slice

__device__ __forceinline__ long long valid_rows(long long R, const long long *r_dev) {
    long long v = r_dev ? *r_dev : R;
    return v < R ? v : R;
}

__device__ __forceinline__ void store8(float *p, const float *x) {
    *(float4 *)p = float4{x[0], x[1], x[2], x[3]};
    *(float4 *)(p + 4) = float4{x[4], x[5], x[6], x[7]};
}
template <typename H>
__device__ __forceinline__ void store8(H *p, const float *x) {
    uint4 v;
    v.x = wfs_pack2<H>(x[0], x[1]);
    v.y = wfs_pack2<H>(x[2], x[3]);
    v.z = wfs_pack2<H>(x[4], x[5]);
    v.w = wfs_pack2<H>(x[6], x[7]);
    *(uint4 *)p = v;
}

// 8 consecutive channels of a row as the registers they travel in (widened to floats only when used): a batch of
// SH_BATCH of them is in flight together
template <typename H> struct Piece { uint4 v; };
template <> struct Piece<float> { float4 a, b; };
template <typename H>
__device__ __forceinline__ Piece<H> load_piece(const H *p) {
    Piece<H> r;
    r.v = *(const uint4 *)p;
    return r;
}
template <>
__device__ __forceinline__ Piece<float> load_piece<float>(const float *p) {
    Piece<float> r;
    r.a = *(const float4 *)p;
    r.b = *(const float4 *)(p + 4);
    return r;
}
template <typename H>
__device__ __forceinline__ void widen(const Piece<H> &q, float *x) {
    wfs_unpack2<H>(q.v.x, x[0], x[1]);
    wfs_unpack2<H>(q.v.y, x[2], x[3]);
    wfs_unpack2<H>(q.v.z, x[4], x[5]);
    wfs_unpack2<H>(q.v.w, x[6], x[7]);
}
template <>
__device__ __forceinline__ void widen<float>(const Piece<float> &q, float *x) {
    x[0] = q.a.x; x[1] = q.a.y; x[2] = q.a.z; x[3] = q.a.w; x[4] = q.b.x; x[5] = q.b.y; x[6] = q.b.z; x[7] = q.b.w;
}
// events whose row loads a thread has in flight together (fp32 rows are twice the registers)
template <typename H> struct ShBatch { static constexpr int n = 8; };
template <> struct ShBatch<float> { static constexpr int n = 4; };

template <typename H, int O>
__global__ void __launch_bounds__(512) k_shead_fwd(const H *__restrict__ X, const unsigned *__restrict__ ticket,
                                                    const int *__restrict__ slot, long long M,
                                                    const long long *__restrict__ m_dev, int B, int V, int C,
                                                    const float *__restrict__ W, float *__restrict__ part) {
    constexpr int SH_BATCH = 8;
    constexpr int NV = SH_BATCH * O;                    // (event, output) sums of one batch
    constexpr int TS = NV | 1;                          // odd stride: the transposing LDS accesses are conflict-free
    extern __shared__ float sh_lds[];
    float *sT = sh_lds + (threadIdx.x >> 6) * (64 * TS);            // per wave: [lane][NV] partial sums of a batch
    float *sP = sh_lds + (blockDim.x >> 6) * (64 * TS);             // [channel group = wave][event of the slice][output]
    const int lane = threadIdx.x & 63, cg = threadIdx.x >> 6, ncg = blockDim.x >> 6;
    const int tile = blockIdx.x, slice = blockIdx.y;
    const int cell = tile * SH_CELLS + lane;
    const bool in = cell < V;
    const int cc = in ? cell : V - 1;
    const long long Mv = valid_rows(M, m_dev);
    // every load of the thread is issued before the first use: weights (coalesced over the cells), the slice's map
    // entries, then the rows they name
    float w[O][8];
#pragma unroll
    for (int o = 0; o < O; ++o)
#pragma unroll
        for (int j = 0; j < 8; ++j) w[o][j] = W[((long long)o * C + cg * 8 + j) * V + cc];
    int row[SH_EVENTS];
#pragma unroll
    for (int e = 0; e < SH_EVENTS; ++e) {
        const int b = slice * SH_EVENTS + e;
        const long long at = (long long)(b < B ? b : B - 1) * V + cc;
        const unsigned t = ticket[at];
        const int r = slot[at];
        row[e] = (in && b < B && t != 0xFFFFFFFFu && r >= 0 && r < Mv) ? r : -1;
    }
#pragma unroll
    for (int e0 = 0; e0 < SH_EVENTS; e0 += SH_BATCH) {
        // the batch's rows: unconditional loads (a missing row reads row 0 and is masked afterwards) -- one memory round
        // trip per batch instead of one per event
        Piece<H> q[SH_BATCH];
#pragma unroll
        for (int i = 0; i < SH_BATCH; ++i) q[i] = load_piece<H>(X + (long long)(row[e0 + i] >= 0 ? row[e0 + i] : 0) * C + cg * 8);
        // this lane's products, then ONE transposing pass through LDS sums each (event, output) over the wave's 64 cells
        // (a shuffle tree per value would be 6 dependent cross-lane steps x 48 values)
#pragma unroll
        for (int i = 0; i < SH_BATCH; ++i) {
            float x[8];
            widen<H>(q[i], x);
            const bool act = row[e0 + i] >= 0;
#pragma unroll
            for (int o = 0; o < O; ++o) {
                float p = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) p = fmaf(x[j], w[o][j], p);
                sT[lane * TS + i * O + o] = act ? p : 0.f;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < NV) {
            float s = 0.f;
#pragma unroll 16
            for (int l = 0; l < 64; ++l) s += sT[l * TS + lane];
            sP[(cg * SH_EVENTS + e0) * O + lane] = s;              // lane = i * O + o
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (threadIdx.x < SH_EVENTS * O) {
        const int e = threadIdx.x / O;
        const int b = slice * SH_EVENTS + e;
        float s = 0.f;
        for (int g = 0; g < ncg; ++g) s += sP[g * SH_EVENTS * O + threadIdx.x];
        if (b < B) part[((long long)tile * B + b) * O + (threadIdx.x % O)] = s;
    }
}

__global__ void __launch_bounds__(256) k_shead_sum(const float *__restrict__ part, int ntiles, int B, int O,
                                                   const float *__restrict__ bias, float *__restrict__ Y) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * O) return;
    float s = bias ? bias[i % O] : 0.f;
    // the loads of a group of 8 tiles are issued together (a plain loop waits for each before it adds), summed in order
    for (int t0 = 0; t0 < ntiles; t0 += 8) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = part[(long long)(t0 + t < ntiles ? t0 + t : ntiles - 1) * B * O + i];
#pragma unroll
        for (int t = 0; t < 8; ++t) s += t0 + t < ntiles ? v[t] : 0.f;
    }
    Y[i] = s;
}

template <typename H, int O>
__global__ void __launch_bounds__(512) k_shead_bwd(const H *__restrict__ X, const float *__restrict__ G,
                                                    const unsigned *__restrict__ ticket, const int *__restrict__ slot,
                                                    long long M, const long long *__restrict__ m_dev, int B, int V, int C,
                                                    const float *__restrict__ W, H *__restrict__ dX,
                                                    float *__restrict__ part, float *__restrict__ dB) {
    const int lane = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int tile = blockIdx.x, slice = blockIdx.y;
    const int cell = tile * SH_CELLS + lane;
    const bool in = cell < V;
    const int cc = in ? cell : V - 1;
    const long long Mv = valid_rows(M, m_dev);
    float w[O][8];
    if (dX) {
#pragma unroll
        for (int o = 0; o < O; ++o)
#pragma unroll
            for (int j = 0; j < 8; ++j) w[o][j] = W[((long long)o * C + cg * 8 + j) * V + cc];
    }
    int row[SH_EVENTS];
#pragma unroll
    for (int e = 0; e < SH_EVENTS; ++e) {
        const int b = slice * SH_EVENTS + e;
        const long long at = (long long)(b < B ? b : B - 1) * V + cc;
        const unsigned t = ticket[at];
        const int r = slot[at];
        row[e] = (in && b < B && t != 0xFFFFFFFFu && r >= 0 && r < Mv) ? r : -1;
    }
    float acc[O][8];
#pragma unroll
    for (int o = 0; o < O; ++o)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[o][j] = 0.f;
    constexpr int SH_BATCH = ShBatch<H>::n;
#pragma unroll
    for (int e0 = 0; e0 < SH_EVENTS; e0 += SH_BATCH) {
        Piece<H> q[SH_BATCH];
        float g[SH_BATCH][O];
#pragma unroll
        for (int i = 0; i < SH_BATCH; ++i) {
            if (part) q[i] = load_piece<H>(X + (long long)(row[e0 + i] >= 0 ? row[e0 + i] : 0) * C + cg * 8);
            const int b = slice * SH_EVENTS + e0 + i;
#pragma unroll
            for (int o = 0; o < O; ++o) g[i][o] = G[(long long)(b < B ? b : B - 1) * O + o];
        }
#pragma unroll
        for (int i = 0; i < SH_BATCH; ++i) {
            const int e = e0 + i;
            const bool act = row[e] >= 0;
            if (part) {
                float x[8];
                widen<H>(q[i], x);
#pragma unroll
                for (int o = 0; o < O; ++o) {
                    const float go = act ? g[i][o] : 0.f;            // a missing row adds 0 (its registers hold row 0)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[o][j] = fmaf(go, x[j], acc[o][j]);
                }
            }
            if (dX && act) {
                float d[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float t = 0.f;
#pragma unroll
                    for (int o = 0; o < O; ++o) t = fmaf(g[i][o], w[o][j], t);
                    d[j] = t;
                }
                store8(dX + (long long)row[e] * C + cg * 8, d);
            }
        }
    }
    if (part && in) {
        float *dst = part + (long long)slice * O * C * V;
#pragma unroll
        for (int o = 0; o < O; ++o)
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[((long long)o * C + cg * 8 + j) * V + cell] = acc[o][j];
    }
    if (dB && tile == 0 && slice == 0 && cg == 0) {
        // db[o] = sum over the events of g[b][o], lanes over the events, fixed order
#pragma unroll
        for (int o = 0; o < O; ++o) {
            float s = 0.f;
            for (int b = lane; b < B; b += 64) s += G[(long long)b * O + o];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
            if (lane == 0) dB[o] = s;
        }
    }
}

// LDS of k_shead_fwd: per wave the [64][8 O | 1] transposing buffer, then the [waves][events][O] sums
inline size_t fwd_lds(int waves, int O) { return ((size_t)waves * 64 * ((8 * O) | 1) + (size_t)waves * SH_EVENTS * O) * sizeof(float); }
inline int sh_tiles(long long V) { return (int)wfs_cdiv(V, SH_CELLS); }
inline int sh_slices(int B) { return (int)wfs_cdiv(B, SH_EVENTS); }

}  // namespace

extern "C" int wfs_sparse_head_ok(int32_t batch, int64_t V, int32_t C, int32_t O, int32_t dtype) {
    return batch >= 1 && batch <= 65535 * SH_EVENTS && V >= 1 && V < (1ll << 24) && C >= 8 && C <= 64 && C % 8 == 0 &&
           O >= 1 && O <= 4 && wfs_dtype_ok(dtype) && (long long)O * C * V < (1ll << 31) && (long long)batch * V < (1ll << 31);
}

// forward: the per-tile partial sums [tiles][batch][O]; backward: the per-slice dW partials [slices][O][C * V]
extern "C" size_t wfs_sparse_head_workspace_bytes(int32_t batch, int64_t V, int32_t C, int32_t O) {
    const size_t fwd = (size_t)sh_tiles(V) * (size_t)batch * O * sizeof(float);
    const size_t bwd = (size_t)sh_slices(batch) * (size_t)O * C * (size_t)V * sizeof(float);
    return fwd > bwd ? fwd : bwd;
}

#define WFS_SH_DISPATCH(O, CALL)                      \
    switch (O) {                                      \
        case 1: { constexpr int OO = 1; CALL; } break; \
        case 2: { constexpr int OO = 2; CALL; } break; \
        case 3: { constexpr int OO = 3; CALL; } break; \
        default: { constexpr int OO = 4; CALL; } break; \
    }

extern "C" int wfs_sparse_head_fwd(const void *X, const uint32_t *ticket, const int32_t *slot_id, int64_t M,
                                   const int64_t *m_dev, int32_t batch, int64_t V, int32_t C, const float *W,
                                   const float *bias, int32_t O, float *Y, int32_t dtype, void *workspace,
                                   size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(wfs_sparse_head_ok(batch, V, C, O, dtype), WFS_EINVAL, "wfs_sparse_head_fwd: shape not covered (wfs_sparse_head_ok)");
    WFS_REQUIRE(X && ticket && slot_id && W && Y && workspace, WFS_EINVAL, "NULL device pointer");
    WFS_REQUIRE(workspace_bytes >= wfs_sparse_head_workspace_bytes(batch, V, C, O), WFS_EWORKSPACE, "workspace too small");
    WFS_REQUIRE(M >= 0, WFS_EINVAL, "M");
    const dim3 grid((unsigned)sh_tiles(V), (unsigned)sh_slices(batch)), block(64 * (C / 8));
    float *part = (float *)workspace;
    const long long *md = (const long long *)m_dev;
#define WFS_SHF(T)                                                                                                      \
    WFS_SH_DISPATCH(O, (k_shead_fwd<T, OO><<<grid, block, fwd_lds(C / 8, OO), stream>>>((const T *)X, ticket, slot_id, M, md, \
                                                                                         batch, (int)V, C, W, part)))
    if (dtype == WFS_F32) { WFS_SHF(float); } else if (dtype == WFS_BF16) { WFS_SHF(wfs_bf16); } else { WFS_SHF(wfs_f16); }
#undef WFS_SHF
    WFS_LAUNCH_CHECK();
    k_shead_sum<<<dim3((unsigned)wfs_cdiv((long long)batch * O, 256)), dim3(256), 0, stream>>>(part, sh_tiles(V), batch, O,
                                                                                               bias, Y);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

extern "C" int wfs_sparse_head_bwd(const void *X, const float *G, const uint32_t *ticket, const int32_t *slot_id,
                                   int64_t M, const int64_t *m_dev, int32_t batch, int64_t V, int32_t C, const float *W,
                                   int32_t O, void *dX, float *dW, float *dB, int32_t dtype, void *workspace,
                                   size_t workspace_bytes, wfs_dw_job *defer, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(wfs_sparse_head_ok(batch, V, C, O, dtype), WFS_EINVAL, "wfs_sparse_head_bwd: shape not covered (wfs_sparse_head_ok)");
    WFS_REQUIRE(X && G && ticket && slot_id && W, WFS_EINVAL, "NULL device pointer");
    WFS_REQUIRE(!dB || dW, WFS_EINVAL, "dB comes with dW");
    if (defer) *defer = wfs_dw_job{nullptr, 0, 0, 0, 0, 0, 0, nullptr};
    if (!dX && !dW) return WFS_OK;
    float *part = nullptr;
    if (dW) {
        WFS_REQUIRE(workspace && workspace_bytes >= wfs_sparse_head_workspace_bytes(batch, V, C, O), WFS_EWORKSPACE,
                    "workspace too small");
        part = (float *)workspace;
    }
    const dim3 grid((unsigned)sh_tiles(V), (unsigned)sh_slices(batch)), block(64 * (C / 8));
    const long long *md = (const long long *)m_dev;
#define WFS_SHB(T)                                                                                                      \
    WFS_SH_DISPATCH(O, (k_shead_bwd<T, OO><<<grid, block, 0, stream>>>((const T *)X, G, ticket, slot_id, M, md, batch, (int)V, \
                                                                        C, W, (T *)dX, part, dB)))
    if (dtype == WFS_F32) { WFS_SHB(float); } else if (dtype == WFS_BF16) { WFS_SHB(wfs_bf16); } else { WFS_SHB(wfs_f16); }
#undef WFS_SHB
    WFS_LAUNCH_CHECK();
    if (!dW) return WFS_OK;
    const long long per = (long long)O * C * V;
    const wfs_dw_job job = {part, sh_slices(batch), per, 1, 1, 1, 0, dW};
    if (defer) {
        *defer = job;
        return WFS_OK;
    }
    return wfs_launch_dw_jobs(&job, 1, stream);
}
