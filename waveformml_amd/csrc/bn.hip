// bn.hip -- BatchNorm1d (+ ReLU) over the active rows of a sparse tensor, forward and backward.
//
// The reference applies plain nn.BatchNorm1d / nn.ReLU modules to SparseConvTensor.features inside
// spconv.SparseSequential (reference src/models/SPConvBlocks.py:505-508): batch
// statistics over the N ACTIVE voxels only, per rank (no SyncBN).  Same arithmetic here, in two launches per
// direction: a column reduction into <= 256 per-block partials, and the elementwise pass, whose every block first
// folds those partials in a fixed order (deterministic, no atomics; at the PSD batch sizes a launch costs ~5 us, more
// than re-reading 64 KB of L2-resident partials per block) and whose block 0 publishes the statistics.
//
//   forward   mean_c, var_c (biased) over rows;  y = max(0, gamma*(x-mean)*invstd + beta)   [ReLU optional]
//             running_mean/var updated with momentum (unbiased var), as torch does
//   backward  g = dy * [y > 0];  dbeta = sum g;  dgamma = sum g*xhat;
//             dx = gamma*invstd*(g - dbeta/N - xhat*dgamma/N)              (training)
//             dx = gamma*invstd*g                                           (eval: running statistics)
#include "wfs_common.h"

#include <atomic>
#include <cstdlib>
#include <mutex>

namespace {

constexpr int TB = 256;
constexpr int MAXC = 1024;

// N = capacity (grid sizing); the number of valid rows comes from device memory when n_dev is given
__device__ __forceinline__ long long valid_rows(long long N, const long long *n_dev) {
    long long v = n_dev ? *n_dev : N;
    return v < N ? v : N;
}

// thread layout of a block: (row slot, channel group of VEC channels); VEC = 4 when C % 4 == 0, else 1
template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T *p, float *out) {
    if constexpr (VEC == 4 && sizeof(T) == 4) {
        float4 v = *reinterpret_cast<const float4 *>(p);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else if constexpr (VEC == 4) {
        uint2 v = *reinterpret_cast<const uint2 *>(p);
        wfs_unpack2<T>(v.x, out[0], out[1]);
        wfs_unpack2<T>(v.y, out[2], out[3]);
    } else {
        out[0] = wfs_ld(p);
    }
}
template <typename T, int VEC>
__device__ __forceinline__ void store_vec(T *p, const float *in) {
    if constexpr (VEC == 4 && sizeof(T) == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(in[0], in[1], in[2], in[3]);
    } else if constexpr (VEC == 4) {
        uint2 v;
        v.x = wfs_pack2<T>(in[0], in[1]);
        v.y = wfs_pack2<T>(in[2], in[3]);
        *reinterpret_cast<uint2 *>(p) = v;
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) wfs_st(p + i, in[i]);
    }
}

// partial[block][2][C]: column sums of (a, b) where
//   MODE 0 (forward stats):  a = d, b = d*d with d = x - x[row 0] (shifted sums: no cancellation in the variance)
//   MODE 1 (backward):       a = g, b = g*xhat with g = dy*[gamma*xhat+beta > 0] (relu), xhat = (x-mean)*invstd
template <typename T, int VEC, int MODE>
__global__ void __launch_bounds__(TB) k_bn_reduce(const T *__restrict__ X, const T *__restrict__ dY, long long Ncap,
                                                  const long long *__restrict__ n_dev, int C, long long ld, long long rows_per_block, const float *__restrict__ mean,
                                                  const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                  const float *__restrict__ beta, int relu,
                                                  float *__restrict__ partial) {
    __shared__ float red[2][TB][VEC];
    const long long N = valid_rows(Ncap, n_dev);
    const int groups = C / VEC, slots = TB / groups;
    const int grp = threadIdx.x % groups, slot = threadIdx.x / groups;
    const int c0 = grp * VEC;
    float sa[VEC], sb[VEC], m[VEC], is[VEC], ga[VEC], be[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) sa[i] = sb[i] = m[i] = is[i] = ga[i] = be[i] = 0.f;
    const bool active = slot < slots;
    if (active) {
        if (MODE == 0) {
            load_vec<T, VEC>(X + c0, m);                      // the shift: row 0
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                m[i] = mean[c0 + i];
                is[i] = invstd[c0 + i];
                ga[i] = gamma ? gamma[c0 + i] : 1.f;
                be[i] = beta ? beta[c0 + i] : 0.f;
            }
        }
        const long long r_begin = (long long)blockIdx.x * rows_per_block;
        const long long r_end = r_begin + rows_per_block < N ? r_begin + rows_per_block : N;
#pragma unroll 4
        for (long long r = r_begin + slot; r < r_end; r += slots) {
            float x[VEC], g[VEC];
            load_vec<T, VEC>(X + r * ld + c0, x);
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float d = x[i] - m[i];
                    sa[i] += d;
                    sb[i] = fmaf(d, d, sb[i]);
                }
            } else {
                load_vec<T, VEC>(dY + r * ld + c0, g);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float xh = (x[i] - m[i]) * is[i];
                    float gi = g[i];
                    if (relu && !(fmaf(ga[i], xh, be[i]) > 0.f)) gi = 0.f;
                    sa[i] += gi;
                    sb[i] = fmaf(gi, xh, sb[i]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        red[0][threadIdx.x][i] = sa[i];
        red[1][threadIdx.x][i] = sb[i];
    }
    __syncthreads();
    if (active && slot == 0) {                       // fold the row slots in a fixed order
        for (int s = 1; s < slots; ++s)
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                sa[i] += red[0][s * groups + grp][i];
                sb[i] += red[1][s * groups + grp][i];
            }
        float *p = partial + (long long)blockIdx.x * 2 * C;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            p[c0 + i] = sa[i];
            p[C + c0 + i] = sb[i];
        }
    }
}

// Sums of the per-block partials [nblk][2][C] in a fixed order, computed by EVERY block of the elementwise kernels in
// their prologue (the partials are a few KB and L2-resident; a separate fold launch costs more than this):
// S = TB / Cp slices x Cp columns, Cp = C rounded up to a power of two.  Slice s owns partials s, s + S, ...; the
// reduce kernel launches at most FOLD_PER * S blocks, so a thread has at most FOLD_PER partials and issues ALL its
// loads before the first add (one memory round trip); slices are then added in slice order.
constexpr int FOLD_PER = 32;
__device__ __forceinline__ int fold_cp(int C) {
    int Cp = 1;
    while (Cp < C && Cp < TB) Cp <<= 1;
    return Cp;
}
__device__ __forceinline__ void fold_partials(const float *__restrict__ partial, int nblk, int C, float *sSlice,
                                              float *sA, float *sB) {
    const int Cp = fold_cp(C);
    const int S = TB / Cp;                                  // 1 when C >= TB
    const int col = threadIdx.x % Cp, sl = threadIdx.x / Cp;
    const long long st = 2ll * C;
    for (int cbase = 0; cbase < C; cbase += Cp) {          // one pass unless C > TB
        const int c = cbase + col;
        const int cc = c < C ? c : 0;
        float av[FOLD_PER], bv[FOLD_PER];
#pragma unroll
        for (int i = 0; i < FOLD_PER; ++i) {
            const int p = sl + i * S;
            const float *q = partial + (long long)(p < nblk ? p : 0) * st + cc;
            av[i] = q[0];
            bv[i] = q[C];
        }
        float a = 0.f, bb = 0.f;
#pragma unroll
        for (int i = 0; i < FOLD_PER; ++i) {
            const bool ok = (sl + i * S < nblk) & (c < C);
            a += ok ? av[i] : 0.f;
            bb += ok ? bv[i] : 0.f;
        }
        sSlice[threadIdx.x] = a;
        sSlice[TB + threadIdx.x] = bb;
        __syncthreads();
        if (sl == 0 && c < C) {
            float ta = 0.f, tb = 0.f;
            for (int q = 0; q < S; ++q) {
                ta += sSlice[q * Cp + col];
                tb += sSlice[TB + q * Cp + col];
            }
            sA[c] = ta;
            sB[c] = tb;
        }
        __syncthreads();
    }
}

// y = [relu](gamma * (x - mean) * invstd + beta).  Statistics: folded from `partial` (training, the reduce kernel's
// shifted sums; block 0 also publishes save_mean / save_invstd and updates the running statistics), or given in
// save_mean / save_invstd (training with known statistics: partial == NULL), or the running ones (eval).
template <typename T, int VEC>
__global__ void __launch_bounds__(TB) k_bn_apply(const T *__restrict__ X, long long Ncap,
                                                 const long long *__restrict__ n_dev, int C, long long ld, long long rows_per_block,
                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                 float *__restrict__ running_mean, float *__restrict__ running_var,
                                                 long long *__restrict__ batches_tracked, float momentum, float eps,
                                                 int training, int relu, T *__restrict__ Y,
                                                 float *__restrict__ save_mean, float *__restrict__ save_invstd,
                                                 const float *__restrict__ partial, int nblk) {
    __shared__ float sSlice[2 * TB];
    __shared__ float sA[MAXC], sB[MAXC];
    const long long N = valid_rows(Ncap, n_dev);
    if (partial) {
        fold_partials(partial, nblk, C, sSlice, sA, sB);
        const float n = N > 0 ? (float)N : 1.f;
        for (int c = threadIdx.x; c < C; c += TB) {
            float shift = wfs_ld(X + c);
            float md = sA[c] / n;                       // mean of (x - shift)
            float var = sB[c] / n - md * md;            // biased, what torch normalises with
            var = var > 0.f ? var : 0.f;
            float mean = shift + md, inv = rsqrtf(var + eps);
            sA[c] = mean;
            sB[c] = inv;
            if (blockIdx.x == 0) {
                save_mean[c] = mean;
                save_invstd[c] = inv;
                if (running_mean) {
                    float unbiased = N > 1 ? var * (n / (n - 1.f)) : var;
                    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
                    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
                }
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0 && batches_tracked) *batches_tracked += 1;
        __syncthreads();
    }
    const int groups = C / VEC, slots = TB / groups;
    const int grp = threadIdx.x % groups, slot = threadIdx.x / groups;
    if (slot >= slots) return;
    const int c0 = grp * VEC;
    float m[VEC], is[VEC], ga[VEC], be[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        if (partial) {
            m[i] = sA[c0 + i];
            is[i] = sB[c0 + i];
        } else if (training) {
            m[i] = save_mean[c0 + i];
            is[i] = save_invstd[c0 + i];
        } else {
            m[i] = running_mean[c0 + i];
            is[i] = rsqrtf(running_var[c0 + i] + eps);
            if (blockIdx.x == 0 && slot == 0) {
                save_mean[c0 + i] = m[i];
                save_invstd[c0 + i] = is[i];
            }
        }
        ga[i] = gamma ? gamma[c0 + i] : 1.f;
        be[i] = beta ? beta[c0 + i] : 0.f;
    }
    const long long r_begin = (long long)blockIdx.x * rows_per_block;
    const long long r_end = r_begin + rows_per_block < N ? r_begin + rows_per_block : N;
#pragma unroll 4
    for (long long r = r_begin + slot; r < r_end; r += slots) {
        float x[VEC], y[VEC];
        load_vec<T, VEC>(X + r * ld + c0, x);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float v = fmaf(ga[i], (x[i] - m[i]) * is[i], be[i]);      // same expression as the backward's mask
            y[i] = (relu && !(v > 0.f)) ? 0.f : v;
        }
        store_vec<T, VEC>(Y + r * ld + c0, y);
    }
}

// dx; the sums (sum g, sum g*xhat) are folded from the reduce kernel's partials in the prologue, block 0 publishes
// them as dbeta / dgamma.
template <typename T, int VEC>
__global__ void __launch_bounds__(TB) k_bn_bwd_apply(const T *__restrict__ X, const T *__restrict__ dY,
                                                     long long Ncap, const long long *__restrict__ n_dev, int C, long long ld,
                                                     long long rows_per_block, const float *__restrict__ partial,
                                                     int nblk,
                                                     const float *__restrict__ mean, const float *__restrict__ invstd,
                                                     const float *__restrict__ gamma, const float *__restrict__ beta,
                                                     int training, int relu, T *__restrict__ dX,
                                                     float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ float sSlice[2 * TB];
    __shared__ float sA[MAXC], sB[MAXC];
    const long long N = valid_rows(Ncap, n_dev);
    fold_partials(partial, nblk, C, sSlice, sA, sB);
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += TB) {
            if (dbeta) dbeta[c] = sA[c];
            if (dgamma) dgamma[c] = sB[c];
        }
    }
    const int groups = C / VEC, slots = TB / groups;
    const int grp = threadIdx.x % groups, slot = threadIdx.x / groups;
    if (slot >= slots) return;
    const int c0 = grp * VEC;
    float m[VEC], is[VEC], ga[VEC], be[VEC], k1[VEC], k2[VEC];
    const float invN = N > 0 ? 1.f / (float)N : 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        m[i] = mean[c0 + i];
        is[i] = invstd[c0 + i];
        ga[i] = gamma ? gamma[c0 + i] : 1.f;
        be[i] = beta ? beta[c0 + i] : 0.f;
        k1[i] = training ? sA[c0 + i] * invN : 0.f;
        k2[i] = training ? sB[c0 + i] * invN : 0.f;
    }
    const long long r_begin = (long long)blockIdx.x * rows_per_block;
    const long long r_end = r_begin + rows_per_block < N ? r_begin + rows_per_block : N;
#pragma unroll 4
    for (long long r = r_begin + slot; r < r_end; r += slots) {
        float x[VEC], g[VEC], o[VEC];
        load_vec<T, VEC>(X + r * ld + c0, x);
        load_vec<T, VEC>(dY + r * ld + c0, g);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float xh = (x[i] - m[i]) * is[i];
            float gi = g[i];
            if (relu && !(fmaf(ga[i], xh, be[i]) > 0.f)) gi = 0.f;
            o[i] = ga[i] * is[i] * (gi - k1[i] - xh * k2[i]);
        }
        store_vec<T, VEC>(dX + r * ld + c0, o);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Register-resident variants for batches whose rows fit the chip's register files (the PSD sizes: ~10^5 x 32).
// A kernel node of a replayed graph costs ~1.7 us empty and ~2.8 us for a 5.5 MB elementwise pass
// (tools/exp/launch_floor.hip); what made the loop kernels above take 6-10 us is DEPENDENT memory round trips at ~1 us
// each: valid count -> loop bound -> row loads -> statistics -> ...  Here every thread owns PER rows
// (first + i * stride) and issues ALL its loads -- rows (bounded by the capacity, not by the count), count, shift,
// partials, affine parameters -- back to back at the top: one round trip, then arithmetic and stores.
constexpr int RR_BLOCKS = 256;
// Timing knock-outs (tools/exp/knock_bn.sh, profiles/r02_bn_knockouts.txt): -DWFS_BN_KNOCK=bits builds a library whose
// register-resident BatchNorm kernels skip a phase -- 1 the fold of the partials, 2 the row loads, 4 the row stores,
// 8 the block reduction + partial store of the reduce kernel.  Results are wrong by construction; 0 = nothing.
#ifndef WFS_BN_KNOCK
#define WFS_BN_KNOCK 0
#endif

// Block sums of the per-thread (sa, sb)[VEC] of the register-resident kernels; thread = (row slot, channel group),
// groups = C / VEC.  Returns true in the threads (threadIdx.x < groups) that end up holding the block's sums of their
// channel group.  When `groups` divides the wave, the slots a wave holds are added by lane exchanges (xor over the slot
// bits: a fixed tree) and the TB / 64 wave sums through LDS in wave order; otherwise slot 0 adds the slots serially.
template <int VEC>
__device__ __forceinline__ bool rr_block_sums(float *sa, float *sb, float (&red)[2][TB][VEC], int groups, bool active) {
    const bool fast = groups <= 64 && (groups & (groups - 1)) == 0;            // TB is a multiple of 64: all active
    if (fast) {
        for (int off = groups; off < 64; off <<= 1)
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                sa[i] += __shfl_xor(sa[i], off, 64);
                sb[i] += __shfl_xor(sb[i], off, 64);
            }
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane < groups) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                red[0][wv * groups + lane][i] = sa[i];
                red[1][wv * groups + lane][i] = sb[i];
            }
        }
        __syncthreads();
        if (threadIdx.x < groups) {
#pragma unroll
            for (int w = 1; w < TB / 64; ++w)
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    sa[i] += red[0][w * groups + threadIdx.x][i];
                    sb[i] += red[1][w * groups + threadIdx.x][i];
                }
        }
        return threadIdx.x < groups;
    }
    const int slots = TB / groups, grp = threadIdx.x % groups, slot = threadIdx.x / groups;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        red[0][threadIdx.x][i] = sa[i];
        red[1][threadIdx.x][i] = sb[i];
    }
    __syncthreads();
    if (active && slot == 0) {
        for (int q = 1; q < slots; ++q)
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                sa[i] += red[0][q * groups + grp][i];
                sb[i] += red[1][q * groups + grp][i];
            }
    }
    return active && slot == 0;
}

template <typename T>
struct Raw4 {                                     // four 16-bit channels (bf16 / fp16) as loaded
    uint2 v;
    __device__ __forceinline__ void load(const T *p) { v = *reinterpret_cast<const uint2 *>(p); }
    __device__ __forceinline__ void get(float *o) const {
        wfs_unpack2<T>(v.x, o[0], o[1]);
        wfs_unpack2<T>(v.y, o[2], o[3]);
    }
};
template <>
struct Raw4<float> {
    float4 v;
    __device__ __forceinline__ void load(const float *p) { v = *reinterpret_cast<const float4 *>(p); }
    __device__ __forceinline__ void get(float *o) const { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
};

template <typename T, int PER, int MODE>
__global__ void __launch_bounds__(TB) k_bn_reduce_rr(const T *__restrict__ X, const T *__restrict__ dY, long long Ncap,
                                                     const long long *__restrict__ n_dev, int C,
                                                     const float *__restrict__ mean, const float *__restrict__ invstd,
                                                     const float *__restrict__ gamma, const float *__restrict__ beta,
                                                     int relu, float *__restrict__ partial) {
    constexpr int VEC = 4;
    __shared__ float red[2][TB][VEC];
    const int groups = C / VEC, slots = TB / groups;
    const int grp = threadIdx.x % groups, slot = threadIdx.x / groups;
    const bool active = slot < slots;
    const int c0 = grp * VEC;
    const long long stride = (long long)gridDim.x * slots, first = (long long)blockIdx.x * slots + slot;
    Raw4<T> x[PER], g[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const long long r = first + i * stride;
        const long long rc = ((WFS_BN_KNOCK & 2) || !(active && r < Ncap)) ? 0 : r;
        x[i].load(X + rc * C + c0);
        if (MODE == 1) g[i].load(dY + rc * C + c0);
    }
    const long long N = valid_rows(Ncap, n_dev);                 // in flight together with the rows
    float m[VEC], is[VEC], ga[VEC], be[VEC], sa[VEC], sb[VEC];
    if (MODE == 0) {
        load_vec<T, VEC>(X + c0, m);                             // the shift: row 0
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        sa[i] = sb[i] = 0.f;
        if (MODE == 1) {
            m[i] = mean[c0 + i];
            is[i] = invstd[c0 + i];
            ga[i] = gamma ? gamma[c0 + i] : 1.f;
            be[i] = beta ? beta[c0 + i] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const long long r = first + i * stride;
        if (active && r < N) {
            float xv[VEC], gv[VEC];
            x[i].get(xv);
            if (MODE == 1) g[i].get(gv);
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                if (MODE == 0) {
                    float d = xv[q] - m[q];
                    sa[q] += d;
                    sb[q] = fmaf(d, d, sb[q]);
                } else {
                    float xh = (xv[q] - m[q]) * is[q];
                    float gi = gv[q];
                    if (relu && !(fmaf(ga[q], xh, be[q]) > 0.f)) gi = 0.f;
                    sa[q] += gi;
                    sb[q] = fmaf(gi, xh, sb[q]);
                }
            }
        }
    }
    if (WFS_BN_KNOCK & 8) {
        if (sa[0] == 1.2345e30f) partial[0] = sb[0];
        return;
    }
    if (rr_block_sums<VEC>(sa, sb, red, groups, active)) {
        float *p = partial + (long long)blockIdx.x * 2 * C;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            p[c0 + i] = sa[i];
            p[C + c0 + i] = sb[i];
        }
    }
}

template <typename T, int PER>
__global__ void __launch_bounds__(TB) k_bn_apply_rr(const T *__restrict__ X, long long Ncap,
                                                    const long long *__restrict__ n_dev, int C,
                                                    const float *__restrict__ gamma, const float *__restrict__ beta,
                                                    float *__restrict__ running_mean, float *__restrict__ running_var,
                                                    long long *__restrict__ batches_tracked, float momentum, float eps,
                                                    int relu, T *__restrict__ Y, float *__restrict__ save_mean,
                                                    float *__restrict__ save_invstd, const float *__restrict__ partial,
                                                    int nblk) {
    constexpr int VEC = 4;
    __shared__ float sSlice[2 * TB];
    __shared__ float sA[MAXC], sB[MAXC];
    const int groups = C / VEC, slots = TB / groups;
    const int grp = threadIdx.x % groups, slot = threadIdx.x / groups;
    const bool active = slot < slots;
    const int c0 = grp * VEC;
    const long long stride = (long long)gridDim.x * slots, first = (long long)blockIdx.x * slots + slot;
    Raw4<T> x[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const long long r = first + i * stride;
        x[i].load(X + (((WFS_BN_KNOCK & 2) || !(active && r < Ncap)) ? 0 : r) * C + c0);
    }
    const long long N = valid_rows(Ncap, n_dev);
    float ga[VEC], be[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        ga[i] = (gamma && active) ? gamma[c0 + i] : 1.f;
        be[i] = (beta && active) ? beta[c0 + i] : 0.f;
    }
    if (WFS_BN_KNOCK & 1) {
        for (int c = threadIdx.x; c < C; c += TB) sA[c] = sB[c] = 1.f;
        __syncthreads();
    } else {
        fold_partials(partial, nblk, C, sSlice, sA, sB);             // its loads join the ones above
    }
    const float n = N > 0 ? (float)N : 1.f;
    for (int c = threadIdx.x; c < C; c += TB) {
        float mean, var;
        const float shift = wfs_ld(X + c);
        const float md = sA[c] / n;
        var = sB[c] / n - md * md;
        mean = shift + md;
        var = var > 0.f ? var : 0.f;
        float inv = rsqrtf(var + eps);
        sA[c] = mean;
        sB[c] = inv;
        if (blockIdx.x == 0) {
            save_mean[c] = mean;
            save_invstd[c] = inv;
            if (running_mean) {
                float unbiased = N > 1 ? var * (n / (n - 1.f)) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && batches_tracked) *batches_tracked += 1;
    __syncthreads();
    if (!active) return;
    float m[VEC], is[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        m[i] = sA[c0 + i];
        is[i] = sB[c0 + i];
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const long long r = first + i * stride;
        if (r < N) {
            float v[VEC], y[VEC];
            x[i].get(v);
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                float t = fmaf(ga[q], (v[q] - m[q]) * is[q], be[q]);
                y[q] = (relu && !(t > 0.f)) ? 0.f : t;
            }
            if (!(WFS_BN_KNOCK & 4) || y[0] == 1.2345e30f) store_vec<T, VEC>(Y + r * C + c0, y);
        }
    }
}

template <typename T, int PER>
__global__ void __launch_bounds__(TB) k_bn_bwd_apply_rr(const T *__restrict__ X, const T *__restrict__ dY,
                                                        long long Ncap, const long long *__restrict__ n_dev, int C,
                                                        const float *__restrict__ partial, int nblk,
                                                        const float *__restrict__ mean, const float *__restrict__ invstd,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        int training, int relu, T *__restrict__ dX,
                                                        float *__restrict__ dgamma, float *__restrict__ dbeta) {
    constexpr int VEC = 4;
    __shared__ float sSlice[2 * TB];
    __shared__ float sA[MAXC], sB[MAXC];
    const int groups = C / VEC, slots = TB / groups;
    const int grp = threadIdx.x % groups, slot = threadIdx.x / groups;
    const bool active = slot < slots;
    const int c0 = grp * VEC;
    const long long stride = (long long)gridDim.x * slots, first = (long long)blockIdx.x * slots + slot;
    Raw4<T> x[PER], g[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const long long r = first + i * stride;
        const long long rc = ((WFS_BN_KNOCK & 2) || !(active && r < Ncap)) ? 0 : r;
        x[i].load(X + rc * C + c0);
        g[i].load(dY + rc * C + c0);
    }
    const long long N = valid_rows(Ncap, n_dev);
    float m[VEC], is[VEC], ga[VEC], be[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        m[i] = active ? mean[c0 + i] : 0.f;
        is[i] = active ? invstd[c0 + i] : 0.f;
        ga[i] = (gamma && active) ? gamma[c0 + i] : 1.f;
        be[i] = (beta && active) ? beta[c0 + i] : 0.f;
    }
    if (WFS_BN_KNOCK & 1) {
        for (int c = threadIdx.x; c < C; c += TB) sA[c] = sB[c] = 1.f;
        __syncthreads();
    } else {
        fold_partials(partial, nblk, C, sSlice, sA, sB);
    }
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += TB) {
            if (dbeta) dbeta[c] = sA[c];
            if (dgamma) dgamma[c] = sB[c];
        }
    }
    if (!active) return;
    const float invN = N > 0 ? 1.f / (float)N : 0.f;
    float k1[VEC], k2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        k1[i] = training ? sA[c0 + i] * invN : 0.f;
        k2[i] = training ? sB[c0 + i] * invN : 0.f;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const long long r = first + i * stride;
        if (r < N) {
            float xv[VEC], gv[VEC], o[VEC];
            x[i].get(xv);
            g[i].get(gv);
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                float xh = (xv[q] - m[q]) * is[q];
                float gi = gv[q];
                if (relu && !(fmaf(ga[q], xh, be[q]) > 0.f)) gi = 0.f;
                o[q] = ga[q] * is[q] * (gi - k1[q] - xh * k2[q]);
            }
            if (!(WFS_BN_KNOCK & 4) || o[0] == 1.2345e30f) store_vec<T, VEC>(dX + r * C + c0, o);
        }
    }
}

// blocks and rows per thread of the register-resident kernels; 0 = the batch does not fit (use the loop kernels)
int rr_plan(long long N, int C, int max_per, long long *blocks) {
    if (C % 4 != 0 || C / 4 > TB) return 0;
    int Cp = 1;
    while (Cp < C && Cp < TB) Cp <<= 1;
    long long nb = (long long)FOLD_PER * (TB / Cp);
    if (nb > RR_BLOCKS) nb = RR_BLOCKS;
    const long long slots = TB / (C / 4);
    const long long need = wfs_cdiv(N, nb * slots);
    *blocks = nb;
    for (int per : {2, 4, 8, 16})
        if (need <= per && per <= max_per) return per;
    return 0;
}

long long bn_reduce_blocks(long long N, int C) {      // = number of partials every elementwise block folds
    int Cp = 1;
    while (Cp < C && Cp < TB) Cp <<= 1;
    const long long cap = (long long)FOLD_PER * (TB / Cp);      // 256 at C = 32, 32 from C = 129 on
    long long b = wfs_cdiv(N, 64);
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return b;
}
long long bn_apply_blocks(long long N) {
    long long b = wfs_cdiv(N, 64);
    if (b < 1) b = 1;
    if (b > 256) b = 256;                   // one block per CU: each pays the fold prologue once
    return b;
}

}  // namespace


// ------------------------------------------------------------------------------------------------------------------
// Column-block variants for WIDE layers with few rows (the hybrid net, BASELINE configs[4]: 776 ... 4105 rows of 1697 /
// 1021 / 345 channels behind SPConvBlocks.py:505-508).  The row-block kernels above give such a layer one block per
// 256-channel slice and row range -- a handful of blocks walking their rows one dependent load after the other
// (9 - 12 us per launch, 13 launches per pass at 1697 + 1021 + 345 channels).  Here the grid is 2-D: a block owns 64
// channels (one per lane: 128-B row pieces of 16-bit data, any channel count, any alignment) and one chunk of rows on
// 4 row slots, so the layer is a few hundred blocks of <= 8 row steps.  Same arithmetic and the same fixed-order
// sums (per-chunk partials, folded in chunk order by every elementwise block): deterministic, no atomics.
constexpr int BW_COLS = 64, BW_SLOTS = TB / BW_COLS, BW_MAX_CHUNKS = 256;

static inline long long bw_chunks(long long N) {
    long long n = wfs_cdiv(N, 32);
    return n < 1 ? 1 : (n > BW_MAX_CHUNKS ? BW_MAX_CHUNKS : n);
}
static inline bool bw_ok(long long N, int C) { return C >= 128 && (C > MAXC || C % 4 != 0 || N * (long long)C <= (1ll << 23)); }

// block sums over the row slots, slot order fixed; the result lands in the slot-0 threads
__device__ __forceinline__ void bw_slot_sum(float &a, float &b, float (&red)[2][TB]) {
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x < BW_COLS) {
#pragma unroll
        for (int q = 1; q < BW_SLOTS; ++q) {
            a += red[0][q * BW_COLS + threadIdx.x];
            b += red[1][q * BW_COLS + threadIdx.x];
        }
    }
}

// partial[chunk][2][C], MODE as k_bn_reduce
template <typename T, int MODE>
__global__ void __launch_bounds__(TB) k_bnw_reduce(const T *__restrict__ X, const T *__restrict__ dY, long long Ncap,
                                                   const long long *__restrict__ n_dev, int C, long long rows_per_chunk,
                                                   const float *__restrict__ mean, const float *__restrict__ invstd,
                                                   const float *__restrict__ gamma, const float *__restrict__ beta,
                                                   int relu, float *__restrict__ partial) {
    __shared__ float red[2][TB];
    const long long N = valid_rows(Ncap, n_dev);
    const int lane = threadIdx.x & (BW_COLS - 1), slot = threadIdx.x / BW_COLS;
    const int c = blockIdx.x * BW_COLS + lane;
    const bool ok = c < C;
    const int cc = ok ? c : 0;
    float m = 0.f, is = 0.f, ga = 1.f, be = 0.f;
    if (MODE == 0) {
        m = wfs_ld(X + cc);                                   // the shift: row 0
    } else {
        m = mean[cc];
        is = invstd[cc];
        ga = gamma ? gamma[cc] : 1.f;
        be = beta ? beta[cc] : 0.f;
    }
    const long long r_begin = (long long)blockIdx.y * rows_per_chunk;
    const long long r_end = r_begin + rows_per_chunk < N ? r_begin + rows_per_chunk : N;
    float sa = 0.f, sb = 0.f;
#pragma unroll 4
    for (long long r = r_begin + slot; r < r_end; r += BW_SLOTS) {
        const float x = wfs_ld(X + r * C + cc);
        if (MODE == 0) {
            const float d = x - m;
            sa += d;
            sb = fmaf(d, d, sb);
        } else {
            const float xh = (x - m) * is;
            float gi = wfs_ld(dY + r * C + cc);
            if (relu && !(fmaf(ga, xh, be) > 0.f)) gi = 0.f;
            sa += gi;
            sb = fmaf(gi, xh, sb);
        }
    }
    bw_slot_sum(sa, sb, red);
    if (threadIdx.x < BW_COLS && ok) {
        float *p = partial + (long long)blockIdx.y * 2 * C;
        p[c] = sa;
        p[C + c] = sb;
    }
}

// sums of the chunk partials of this block's 64 channels, chunk order fixed (slot s adds chunks s, s + 4, ...; the
// slots are added in slot order); valid in the slot-0 threads, broadcast to the others through `out`
__device__ __forceinline__ void bw_fold(const float *__restrict__ partial, int nch, int C, int c, bool ok,
                                        float (&red)[2][TB], float (&out)[2][BW_COLS], float &a, float &b) {
    const int lane = threadIdx.x & (BW_COLS - 1), slot = threadIdx.x / BW_COLS;
    a = 0.f;
    b = 0.f;
    if (ok)
        for (int p = slot; p < nch; p += BW_SLOTS) {
            a += partial[(long long)p * 2 * C + c];
            b += partial[(long long)p * 2 * C + C + c];
        }
    bw_slot_sum(a, b, red);
    if (threadIdx.x < BW_COLS) {
        out[0][lane] = a;
        out[1][lane] = b;
    }
    __syncthreads();
    a = out[0][lane];
    b = out[1][lane];
}

template <typename T>
__global__ void __launch_bounds__(TB) k_bnw_apply(const T *__restrict__ X, long long Ncap,
                                                  const long long *__restrict__ n_dev, int C, long long rows_per_chunk,
                                                  const float *__restrict__ gamma, const float *__restrict__ beta,
                                                  float *__restrict__ running_mean, float *__restrict__ running_var,
                                                  long long *__restrict__ batches_tracked, float momentum, float eps,
                                                  int training, int relu, T *__restrict__ Y,
                                                  float *__restrict__ save_mean, float *__restrict__ save_invstd,
                                                  const float *__restrict__ partial, int nch) {
    __shared__ float red[2][TB];
    __shared__ float tot[2][BW_COLS];
    const long long N = valid_rows(Ncap, n_dev);
    const int lane = threadIdx.x & (BW_COLS - 1), slot = threadIdx.x / BW_COLS;
    const int c = blockIdx.x * BW_COLS + lane;
    const bool ok = c < C;
    const int cc = ok ? c : 0;
    const bool publish = blockIdx.y == 0 && slot == 0 && ok;
    float m, is;
    if (training) {
        float a, b;
        bw_fold(partial, nch, C, c, ok, red, tot, a, b);
        const float n = N > 0 ? (float)N : 1.f;
        const float shift = wfs_ld(X + cc);
        const float md = a / n;                               // mean of (x - shift)
        float var = b / n - md * md;                          // biased, what torch normalises with
        var = var > 0.f ? var : 0.f;
        m = shift + md;
        is = rsqrtf(var + eps);
        if (publish) {
            save_mean[c] = m;
            save_invstd[c] = is;
            if (running_mean) {
                const float unbiased = N > 1 ? var * (n / (n - 1.f)) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
            }
        }
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && batches_tracked) *batches_tracked += 1;
    } else {
        m = running_mean[cc];
        is = rsqrtf(running_var[cc] + eps);
        if (publish) {
            save_mean[c] = m;
            save_invstd[c] = is;
        }
    }
    const float ga = gamma ? gamma[cc] : 1.f, be = beta ? beta[cc] : 0.f;
    const long long r_begin = (long long)blockIdx.y * rows_per_chunk;
    const long long r_end = r_begin + rows_per_chunk < N ? r_begin + rows_per_chunk : N;
    if (!ok) return;
#pragma unroll 4
    for (long long r = r_begin + slot; r < r_end; r += BW_SLOTS) {
        const float v = fmaf(ga, (wfs_ld(X + r * C + c) - m) * is, be);      // same expression as the backward's mask
        wfs_st(Y + r * C + c, (relu && !(v > 0.f)) ? 0.f : v);
    }
}

template <typename T>
__global__ void __launch_bounds__(TB) k_bnw_bwd_apply(const T *__restrict__ X, const T *__restrict__ dY, long long Ncap,
                                                      const long long *__restrict__ n_dev, int C,
                                                      long long rows_per_chunk, const float *__restrict__ partial,
                                                      int nch, const float *__restrict__ mean,
                                                      const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, int training, int relu,
                                                      T *__restrict__ dX, float *__restrict__ dgamma,
                                                      float *__restrict__ dbeta) {
    __shared__ float red[2][TB];
    __shared__ float tot[2][BW_COLS];
    const long long N = valid_rows(Ncap, n_dev);
    const int lane = threadIdx.x & (BW_COLS - 1), slot = threadIdx.x / BW_COLS;
    const int c = blockIdx.x * BW_COLS + lane;
    const bool ok = c < C;
    float a, b;
    bw_fold(partial, nch, C, c, ok, red, tot, a, b);
    if (blockIdx.y == 0 && slot == 0 && ok) {
        if (dbeta) dbeta[c] = a;
        if (dgamma) dgamma[c] = b;
    }
    if (!ok) return;
    const float invN = N > 0 ? 1.f / (float)N : 0.f;
    const float m = mean[c], is = invstd[c], ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    const float k1 = training ? a * invN : 0.f, k2 = training ? b * invN : 0.f;
    const long long r_begin = (long long)blockIdx.y * rows_per_chunk;
    const long long r_end = r_begin + rows_per_chunk < N ? r_begin + rows_per_chunk : N;
#pragma unroll 4
    for (long long r = r_begin + slot; r < r_end; r += BW_SLOTS) {
        const float xh = (wfs_ld(X + r * C + c) - m) * is;
        float gi = wfs_ld(dY + r * C + c);
        if (relu && !(fmaf(ga, xh, be) > 0.f)) gi = 0.f;
        wfs_st(dX + r * C + c, ga * is * (gi - k1 - xh * k2));
    }
}

template <typename T>
static int bw_fwd(const void *X, long long N, int C, const float *gamma, const float *beta, float *running_mean,
                  float *running_var, int64_t *num_batches_tracked, float momentum, float eps, int training, int relu,
                  void *Y, float *save_mean, float *save_invstd, float *partial, const long long *n_dev,
                  hipStream_t stream) {
    const long long nch = bw_chunks(N), rpc = wfs_cdiv(wfs_cdiv(N, nch), BW_SLOTS) * BW_SLOTS;
    const dim3 grid((unsigned)wfs_cdiv(C, BW_COLS), (unsigned)nch), block(TB);
    if (training)
        k_bnw_reduce<T, 0><<<grid, block, 0, stream>>>((const T *)X, nullptr, N, n_dev, C, rpc, nullptr, nullptr, nullptr,
                                                       nullptr, 0, partial);
    k_bnw_apply<T><<<grid, block, 0, stream>>>((const T *)X, N, n_dev, C, rpc, gamma, beta, running_mean, running_var,
                                               (long long *)num_batches_tracked, momentum, eps, training, relu, (T *)Y,
                                               save_mean, save_invstd, partial, (int)nch);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

template <typename T>
static int bw_bwd(const void *X, const void *dY, long long N, int C, const float *gamma, const float *beta,
                  const float *save_mean, const float *save_invstd, int training, int relu, void *dX, float *dgamma,
                  float *dbeta, float *partial, const long long *n_dev, hipStream_t stream) {
    const long long nch = bw_chunks(N), rpc = wfs_cdiv(wfs_cdiv(N, nch), BW_SLOTS) * BW_SLOTS;
    const dim3 grid((unsigned)wfs_cdiv(C, BW_COLS), (unsigned)nch), block(TB);
    k_bnw_reduce<T, 1><<<grid, block, 0, stream>>>((const T *)X, (const T *)dY, N, n_dev, C, rpc, save_mean, save_invstd,
                                                   gamma, beta, relu, partial);
    k_bnw_bwd_apply<T><<<grid, block, 0, stream>>>((const T *)X, (const T *)dY, N, n_dev, C, rpc, partial, (int)nch,
                                                   save_mean, save_invstd, gamma, beta, training, relu, (T *)dX, dgamma,
                                                   dbeta);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

extern "C" size_t wfs_bn_workspace_bytes(int64_t N, int32_t C) {
    size_t nb = (size_t)bn_reduce_blocks(N, C);
    if (nb < (size_t)RR_BLOCKS) nb = RR_BLOCKS;
    return (nb + 1) * 2 * C * sizeof(float);
}

// C channels starting at X / Y (row stride ld elements): one slice of a wide layer, or the whole layer (ld == C)
static int bn_fwd_slice(const void *X, int64_t N, int32_t C, long long ld, const float *gamma, const float *beta,
                        float *running_mean, float *running_var, int64_t *num_batches_tracked, float momentum,
                        float eps, int32_t training, int32_t relu, void *Y, float *save_mean, float *save_invstd,
                        void *workspace, size_t workspace_bytes, int32_t dtype, const long long *n_dev,
                        hipStream_t stream) {
    const bool vec4 = C % 4 == 0 && ld % 4 == 0;       // vector loads need 4-element aligned rows
    WFS_REQUIRE(C >= 1 && C <= MAXC && (vec4 ? C / 4 : C) <= TB, WFS_EINVAL, "unsupported channel count %d", C);
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    WFS_REQUIRE(training || (running_mean && running_var), WFS_EINVAL, "eval mode needs running statistics");
    if (N == 0) return WFS_OK;
    WFS_REQUIRE(X && Y && save_mean && save_invstd && workspace, WFS_EINVAL, "NULL device pointer");
    const long long nblk = bn_reduce_blocks(N, C), nblk_a = bn_apply_blocks(N);
    WFS_REQUIRE(workspace_bytes >= wfs_bn_workspace_bytes(N, C), WFS_EWORKSPACE, "workspace too small");
    const long long rpb = wfs_cdiv(N, nblk), rpb_a = wfs_cdiv(N, nblk_a);
    float *partial = (float *)workspace;
    dim3 grid((unsigned)nblk), grid_a((unsigned)nblk_a), block(TB);
    if (training && ld == C) {
        long long rb = 0;
        const int per = rr_plan(N, C, 16, &rb);
        if (per) {
            const dim3 g2((unsigned)rb);
#define WFS_BN_FWD_RR(T, PER)                                                                                          \
    do {                                                                                                               \
        k_bn_reduce_rr<T, PER, 0><<<g2, block, 0, stream>>>((const T *)X, nullptr, N, n_dev, C, nullptr, nullptr,      \
                                                             nullptr, nullptr, 0, partial);                            \
        k_bn_apply_rr<T, PER><<<g2, block, 0, stream>>>(                                                               \
            (const T *)X, N, n_dev, C, gamma, beta, running_mean, running_var, (long long *)num_batches_tracked,       \
            momentum, eps, relu, (T *)Y, save_mean, save_invstd, partial, (int)rb);                                    \
    } while (0)
#define WFS_BN_FWD_RR_T(T)                                                                                             \
    if (per == 2) WFS_BN_FWD_RR(T, 2); else if (per == 4) WFS_BN_FWD_RR(T, 4); else if (per == 8) WFS_BN_FWD_RR(T, 8);   \
    else WFS_BN_FWD_RR(T, 16)
            if (dtype == WFS_F32) { WFS_BN_FWD_RR_T(float); } else if (dtype == WFS_BF16) { WFS_BN_FWD_RR_T(wfs_bf16); } else { WFS_BN_FWD_RR_T(wfs_f16); }
#undef WFS_BN_FWD_RR_T
#undef WFS_BN_FWD_RR
            WFS_LAUNCH_CHECK();
            return WFS_OK;
        }
    }
#define WFS_BN_FWD(T, VEC)                                                                                          \
    do {                                                                                                            \
        if (training)                                                                                               \
            k_bn_reduce<T, VEC, 0><<<grid, block, 0, stream>>>((const T *)X, nullptr, N, n_dev, C, ld, rpb, nullptr, \
                                                                nullptr, nullptr, nullptr, 0, partial);             \
        k_bn_apply<T, VEC><<<grid_a, block, 0, stream>>>(                                                           \
            (const T *)X, N, n_dev, C, ld, rpb_a, gamma, beta, running_mean, running_var,                           \
            (long long *)num_batches_tracked, momentum, eps, training, relu, (T *)Y, save_mean, save_invstd,        \
            training ? partial : nullptr, (int)nblk);                                                               \
    } while (0)
    if (dtype == WFS_F32) {
        if (vec4) WFS_BN_FWD(float, 4); else WFS_BN_FWD(float, 1);
    } else if (dtype == WFS_BF16) {
        if (vec4) WFS_BN_FWD(wfs_bf16, 4); else WFS_BN_FWD(wfs_bf16, 1);
    } else {
        if (vec4) WFS_BN_FWD(wfs_f16, 4); else WFS_BN_FWD(wfs_f16, 1);
    }
#undef WFS_BN_FWD
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

// layers wider than one block's reach (MAXC channels; the hybrid net's first layers have 2 T = 2048) run slice by slice
static inline size_t elem_bytes(int dtype) { return dtype == WFS_F32 ? 4 : 2; }

extern "C" int wfs_bn_relu_fwd(const void *X, int64_t N, int32_t C, const float *gamma, const float *beta,
                               float *running_mean, float *running_var, int64_t *num_batches_tracked, float momentum,
                               float eps, int32_t training, int32_t relu, void *Y, float *save_mean,
                               float *save_invstd, void *workspace, size_t workspace_bytes, int32_t dtype,
                               const int64_t *n_dev_, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const long long *n_dev = (const long long *)n_dev_;
    if (bw_ok(N, C) && N > 0 && wfs_dtype_ok(dtype)) {
        WFS_REQUIRE(training || (running_mean && running_var), WFS_EINVAL, "eval mode needs running statistics");
        WFS_REQUIRE(X && Y && save_mean && save_invstd && workspace, WFS_EINVAL, "NULL device pointer");
        WFS_REQUIRE(workspace_bytes >= wfs_bn_workspace_bytes(N, C), WFS_EWORKSPACE, "workspace too small");
        float *partial = (float *)workspace;
        if (dtype == WFS_F32)
            return bw_fwd<float>(X, N, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                                 training, relu, Y, save_mean, save_invstd, partial, n_dev, stream);
        if (dtype == WFS_BF16)
            return bw_fwd<wfs_bf16>(X, N, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                                    training, relu, Y, save_mean, save_invstd, partial, n_dev, stream);
        return bw_fwd<wfs_f16>(X, N, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                               training, relu, Y, save_mean, save_invstd, partial, n_dev, stream);
    }
    if (C <= (C % 4 == 0 ? MAXC : TB))
        return bn_fwd_slice(X, N, C, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                            training, relu, Y, save_mean, save_invstd, workspace, workspace_bytes, dtype, n_dev, stream);
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    const int width = C % 4 == 0 ? MAXC : TB;          // odd widths: scalar path, one channel per thread
    for (int c0 = 0; c0 < C; c0 += width) {
        const int cs = C - c0 < width ? C - c0 : width;
        const size_t off = (size_t)c0 * elem_bytes(dtype);
        int rc = bn_fwd_slice(X ? (const char *)X + off : nullptr, N, cs, C, gamma ? gamma + c0 : nullptr,
                              beta ? beta + c0 : nullptr, running_mean ? running_mean + c0 : nullptr,
                              running_var ? running_var + c0 : nullptr, c0 == 0 ? num_batches_tracked : nullptr,
                              momentum, eps, training, relu, Y ? (char *)Y + off : nullptr,
                              save_mean ? save_mean + c0 : nullptr, save_invstd ? save_invstd + c0 : nullptr, workspace,
                              workspace_bytes, dtype, n_dev, stream);
        if (rc != WFS_OK) return rc;
    }
    return WFS_OK;
}

static int bn_bwd_slice(const void *X, const void *dY, int64_t N, int32_t C, long long ld, const float *gamma,
                        const float *beta, const float *save_mean, const float *save_invstd, int32_t training,
                        int32_t relu, void *dX, float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes,
                        int32_t dtype, const long long *n_dev, hipStream_t stream) {
    const bool vec4 = C % 4 == 0 && ld % 4 == 0;       // vector loads need 4-element aligned rows
    WFS_REQUIRE(C >= 1 && C <= MAXC && (vec4 ? C / 4 : C) <= TB, WFS_EINVAL, "unsupported channel count %d", C);
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    if (N == 0) {
        if (dgamma) WFS_HIP_CHECK(hipMemsetAsync(dgamma, 0, C * sizeof(float), stream));
        if (dbeta) WFS_HIP_CHECK(hipMemsetAsync(dbeta, 0, C * sizeof(float), stream));
        return WFS_OK;
    }
    WFS_REQUIRE(X && dY && dX && save_mean && save_invstd && workspace, WFS_EINVAL, "NULL device pointer");
    const long long nblk = bn_reduce_blocks(N, C), nblk_a = bn_apply_blocks(N);
    WFS_REQUIRE(workspace_bytes >= wfs_bn_workspace_bytes(N, C), WFS_EWORKSPACE, "workspace too small");
    const long long rpb = wfs_cdiv(N, nblk), rpb_a = wfs_cdiv(N, nblk_a);
    float *partial = (float *)workspace;
    dim3 grid((unsigned)nblk), grid_a((unsigned)nblk_a), block(TB);
    if (ld == C) {
        long long rb = 0;
        const int per = rr_plan(N, C, dtype == WFS_F32 ? 8 : 16, &rb);
        if (per) {
            const dim3 g2((unsigned)rb);
#define WFS_BN_BWD_RR(T, PER)                                                                                          \
    do {                                                                                                               \
        k_bn_reduce_rr<T, PER, 1><<<g2, block, 0, stream>>>((const T *)X, (const T *)dY, N, n_dev, C, save_mean,       \
                                                             save_invstd, gamma, beta, relu, partial);                 \
        k_bn_bwd_apply_rr<T, PER><<<g2, block, 0, stream>>>((const T *)X, (const T *)dY, N, n_dev, C, partial, (int)rb, \
                                                            save_mean, save_invstd, gamma, beta, training, relu,       \
                                                            (T *)dX, dgamma, dbeta);                                   \
    } while (0)
#define WFS_BN_BWD_RR_T(T)                                                                                             \
    if (per == 2) WFS_BN_BWD_RR(T, 2); else if (per == 4) WFS_BN_BWD_RR(T, 4); else if (per == 8) WFS_BN_BWD_RR(T, 8);   \
    else WFS_BN_BWD_RR(T, 16)
            if (dtype == WFS_F32) { WFS_BN_BWD_RR_T(float); } else if (dtype == WFS_BF16) { WFS_BN_BWD_RR_T(wfs_bf16); } else { WFS_BN_BWD_RR_T(wfs_f16); }
#undef WFS_BN_BWD_RR_T
#undef WFS_BN_BWD_RR
            WFS_LAUNCH_CHECK();
            return WFS_OK;
        }
    }
#define WFS_BN_BWD(T, VEC)                                                                                          \
    do {                                                                                                            \
        k_bn_reduce<T, VEC, 1><<<grid, block, 0, stream>>>((const T *)X, (const T *)dY, N, n_dev, C, ld, rpb, save_mean, \
                                                            save_invstd, gamma, beta, relu, partial);               \
        k_bn_bwd_apply<T, VEC><<<grid_a, block, 0, stream>>>((const T *)X, (const T *)dY, N, n_dev, C, ld, rpb_a, partial, \
                                                              (int)nblk, save_mean, save_invstd, gamma, beta,       \
                                                              training, relu, (T *)dX, dgamma, dbeta);              \
    } while (0)
    if (dtype == WFS_F32) {
        if (vec4) WFS_BN_BWD(float, 4); else WFS_BN_BWD(float, 1);
    } else if (dtype == WFS_BF16) {
        if (vec4) WFS_BN_BWD(wfs_bf16, 4); else WFS_BN_BWD(wfs_bf16, 1);
    } else {
        if (vec4) WFS_BN_BWD(wfs_f16, 4); else WFS_BN_BWD(wfs_f16, 1);
    }
#undef WFS_BN_BWD
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

extern "C" int wfs_bn_relu_bwd(const void *X, const void *dY, int64_t N, int32_t C, const float *gamma,
                               const float *beta, const float *save_mean, const float *save_invstd, int32_t training,
                               int32_t relu, void *dX, float *dgamma, float *dbeta, void *workspace,
                               size_t workspace_bytes, int32_t dtype, const int64_t *n_dev_, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const long long *n_dev = (const long long *)n_dev_;
    if (bw_ok(N, C) && N > 0 && wfs_dtype_ok(dtype)) {
        WFS_REQUIRE(X && dY && dX && save_mean && save_invstd && workspace, WFS_EINVAL, "NULL device pointer");
        WFS_REQUIRE(workspace_bytes >= wfs_bn_workspace_bytes(N, C), WFS_EWORKSPACE, "workspace too small");
        float *partial = (float *)workspace;
        if (dtype == WFS_F32)
            return bw_bwd<float>(X, dY, N, C, gamma, beta, save_mean, save_invstd, training, relu, dX, dgamma, dbeta,
                                 partial, n_dev, stream);
        if (dtype == WFS_BF16)
            return bw_bwd<wfs_bf16>(X, dY, N, C, gamma, beta, save_mean, save_invstd, training, relu, dX, dgamma, dbeta,
                                    partial, n_dev, stream);
        return bw_bwd<wfs_f16>(X, dY, N, C, gamma, beta, save_mean, save_invstd, training, relu, dX, dgamma, dbeta,
                               partial, n_dev, stream);
    }
    if (C <= (C % 4 == 0 ? MAXC : TB))
        return bn_bwd_slice(X, dY, N, C, C, gamma, beta, save_mean, save_invstd, training, relu, dX, dgamma, dbeta,
                            workspace, workspace_bytes, dtype, n_dev, stream);
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    const int width = C % 4 == 0 ? MAXC : TB;          // odd widths: scalar path, one channel per thread
    for (int c0 = 0; c0 < C; c0 += width) {
        const int cs = C - c0 < width ? C - c0 : width;
        const size_t off = (size_t)c0 * elem_bytes(dtype);
        int rc = bn_bwd_slice(X ? (const char *)X + off : nullptr, dY ? (const char *)dY + off : nullptr, N, cs, C,
                              gamma ? gamma + c0 : nullptr, beta ? beta + c0 : nullptr,
                              save_mean ? save_mean + c0 : nullptr, save_invstd ? save_invstd + c0 : nullptr, training,
                              relu, dX ? (char *)dX + off : nullptr, dgamma ? dgamma + c0 : nullptr,
                              dbeta ? dbeta + c0 : nullptr, workspace, workspace_bytes, dtype, n_dev, stream);
        if (rc != WFS_OK) return rc;
    }
    return WFS_OK;
}

