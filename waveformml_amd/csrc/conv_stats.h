// conv_stats.h -- BatchNorm statistics taken in the epilogue of the convolution that produces the rows.
//
// The reference runs nn.BatchNorm1d on the conv output as a separate module (src/models/SPConvBlocks.py:505-508); in
// training mode that is a full extra read of [N, C] just for mean / variance.  Here the MFMA conv kernels, which hold
// every output tile in registers anyway, keep per-column running (count, mean, M2) of the values they store:
//   tile   two-pass inside the registers: column sum -> tile mean -> sum of squared deviations from it
//   merge  Chan's parallel update (n, mean, M2) <- (n_a, mean_a, M2_a) + (n_b, mean_b, M2_b); no cancellation,
//          fixed merge order everywhere (tiles in a wave's loop order, waves in wave order, blocks in block order)
//          -> run-to-run identical results, no float atomics
//   fold   one small launch (k_stats_fold) merges the block partials, writes mean / invstd and updates the running
//          statistics exactly as torch does (unbiased variance, momentum).
// Measured alternative (round 1): folding inside the conv kernel by the block that finishes last (ticket atomicAdd +
// fences) cost ~10 us per launch -- every block's tail waits for a device-scope atomic round trip, and a
// __threadfence() per block is buffer_wbl2 + buffer_inv of the XCD's whole L2 -- against ~5 us for this launch.
#pragma once
#include "wfs_common.h"

struct WfsStatsArgs {
    float *part;            // [gridDim.x][2][32]: mean, M2 per block
    float *partn;           // [gridDim.x]: row count per block
    float *save_mean;       // [32] out
    float *save_invstd;     // [32] out
    float *running_mean;    // [32] in/out or NULL
    float *running_var;
    long long *batches_tracked;     // or NULL
    float momentum, eps;
};

struct WfsColStats {
    float n, mean, m2;
};

__device__ __forceinline__ void wfs_chan_merge(WfsColStats &a, float nb, float mb, float m2b) {
    if (nb > 0.f) {
        float n = a.n + nb;
        float d = mb - a.mean;
        float f = nb * __builtin_amdgcn_rcpf(n);          // 1 ulp; an IEEE division is ~10 instructions per merge
        a.mean = fmaf(d, f, a.mean);
        a.m2 = a.m2 + m2b + d * d * (a.n * f);
        a.n = n;
    }
}

// One 32x32 MFMA output tile (C/D layout: lane = (col = lane & 31, h = lane >> 5), register i = row
// (i & 3) + 8 (i >> 2) + 4 h) whose first `nlive` rows are valid; v = the values as stored.
__device__ __forceinline__ void wfs_stats_tile(WfsColStats &st, const float (&v)[16], int nlive, int h) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        s += row < nlive ? v[i] : 0.f;
    }
    s += __shfl_xor(s, 32, 64);
    const float nt = (float)nlive;
    const float mt = s / nt;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        float d = v[i] - mt;
        q = row < nlive ? fmaf(d, d, q) : q;
    }
    q += __shfl_xor(q, 32, 64);
    wfs_chan_merge(st, nt, mt, q);
}

// Cheaper per-tile bookkeeping (what the kernels use): every lane keeps SHIFTED sums of its own 16 rows per tile --
// d = v - shift with shift = the first value the lane ever produced (any value near the column's mean keeps
// sum d^2 - (sum d)^2 / n free of cancellation) -- i.e. two FMAs per value, no shuffles, no divisions inside the tile
// loop.  At the end of the kernel the lane's (n, shift, s1, s2) becomes (n, mean, M2), the two lanes of a column are
// merged (Chan), then the waves (wfs_stats_finish).
struct WfsLaneStats {
    float n, shift, s1, s2;
};

__device__ __forceinline__ void wfs_lane_stats_tile(WfsLaneStats &st, const float (&v)[16], int nlive, int h) {
    if (st.n == 0.f) st.shift = v[0];                 // (a dead row's value is still a finite number near the bias)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        const bool live = row < nlive;
        const float d = v[i] - st.shift;
        st.s1 += live ? d : 0.f;
        st.s2 = live ? fmaf(d, d, st.s2) : st.s2;
        st.n += live ? 1.f : 0.f;
    }
}

__device__ __forceinline__ WfsColStats wfs_lane_stats_close(const WfsLaneStats &st) {
    WfsColStats c = {0.f, 0.f, 0.f};
    if (st.n > 0.f) {
        const float md = st.s1 / st.n;
        c.n = st.n;
        c.mean = st.shift + md;
        const float m2 = st.s2 - st.s1 * md;
        c.m2 = m2 > 0.f ? m2 : 0.f;
    }
    // the column's other lane (h ^ 1) holds the other 16 rows of every tile
    const float on = __shfl_xor(c.n, 32, 64), om = __shfl_xor(c.mean, 32, 64), oq = __shfl_xor(c.m2, 32, 64);
    WfsColStats lo = c, hi = {on, om, oq};
    if ((threadIdx.x & 63) >= 32) {                   // merge in (h = 0, h = 1) order on both lanes: identical results
        lo = hi;
        hi = c;
    }
    wfs_chan_merge(lo, hi.n, hi.mean, hi.m2);
    return lo;
}

// End of the conv kernel, called by EVERY thread of the block (blockDim.x = 64 * nw, nw <= 16): the waves' statistics
// are merged in wave order and stored as this block's partial.  sStat: >= 16 * 65 floats of LDS nobody else touches
// any more.  No atomics, no fences: the partials are consumed by the next launch (k_stats_fold).
__device__ __forceinline__ void wfs_stats_finish(const WfsColStats &st, float *sStat, const WfsStatsArgs &sa) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane < 32) {
        sStat[wid * 65 + lane] = st.mean;
        sStat[wid * 65 + 32 + lane] = st.m2;
        if (lane == 0) sStat[wid * 65 + 64] = st.n;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        WfsColStats b = {0.f, 0.f, 0.f};
        for (int w = 0; w < nw; ++w) wfs_chan_merge(b, sStat[w * 65 + 64], sStat[w * 65 + lane], sStat[w * 65 + 32 + lane]);
        sa.part[((long long)blockIdx.x * 2) * 32 + lane] = b.mean;
        sa.part[((long long)blockIdx.x * 2 + 1) * 32 + lane] = b.m2;
        if (lane == 0) sa.partn[blockIdx.x] = b.n;
    }
}

// One block of 1024 threads = 32 slices x 32 columns: slice sl merges the block partials sl, sl + 32, ... in order
// (four at a time, so the independent loads are in flight together), slices are merged in slice order; then mean,
// invstd and the running statistics exactly as torch updates them (unbiased variance, momentum).
static __global__ void __launch_bounds__(1024) k_stats_fold(WfsStatsArgs sa, int nb) {
    __shared__ float sStat[32 * 65];
    const int t = threadIdx.x, c = t & 31, sl = t >> 5;
    WfsColStats f = {0.f, 0.f, 0.f};
    for (int b0 = sl; b0 < nb; b0 += 4 * 32) {
        float n4[4], m4[4], q4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int b = b0 + u * 32;
            int bc = b < nb ? b : nb - 1;
            float nn = sa.partn[bc];
            m4[u] = sa.part[((long long)bc * 2) * 32 + c];
            q4[u] = sa.part[((long long)bc * 2 + 1) * 32 + c];
            n4[u] = b < nb ? nn : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) wfs_chan_merge(f, n4[u], m4[u], q4[u]);
    }
    sStat[sl * 65 + c] = f.mean;
    sStat[sl * 65 + 32 + c] = f.m2;
    if (c == 0) sStat[sl * 65 + 64] = f.n;
    __syncthreads();
    if (t < 32) {
        WfsColStats g = {0.f, 0.f, 0.f};
        for (int s2 = 0; s2 < 32; ++s2) wfs_chan_merge(g, sStat[s2 * 65 + 64], sStat[s2 * 65 + t], sStat[s2 * 65 + 32 + t]);
        const float n = g.n > 0.f ? g.n : 1.f;
        float var = g.m2 / n;                        // biased, what torch normalises with
        var = var > 0.f ? var : 0.f;
        sa.save_mean[t] = g.mean;
        sa.save_invstd[t] = rsqrtf(var + sa.eps);
        if (sa.running_mean) {
            float unbiased = g.n > 1.f ? g.m2 / (g.n - 1.f) : var;
            sa.running_mean[t] = (1.f - sa.momentum) * sa.running_mean[t] + sa.momentum * g.mean;
            sa.running_var[t] = (1.f - sa.momentum) * sa.running_var[t] + sa.momentum * unbiased;
        }
        if (t == 0 && sa.batches_tracked) *sa.batches_tracked += 1;
    }
}
