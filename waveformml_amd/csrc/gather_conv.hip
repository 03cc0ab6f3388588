// gather_conv.hip -- gather - GEMM - (no) scatter: forward, dX and dW of a sparse convolution.
//
// Replaces torch.ops.spconv.indice_conv / indice_conv_backward of spconv 1.2.1 (SURVEY.md A.4;
// reference call sites src/models/SPConvBlocks.py:75,134,498,803-810).  spconv's Native algo is
// input-stationary: per offset gather -> mm -> scatter-ADD, which on a GPU means float atomics
// (~1.3 TB/s chip-wide on MI355X, 4-5x below plain stores).  Here the rulebook is kept as gather
// tables (rulebook.hip) and every kernel is OUTPUT-stationary: a row of the result is owned by one
// thread group, its <= K source rows are gathered through table[k][row] and contracted with W[k];
// nothing is scattered, nothing is atomically accumulated, results are run-to-run reproducible.
//
// This file holds the shape-generic fp32-accumulate kernels (any Cin/Cout, fp32 or bf16 storage).
// The shape-specialised MFMA kernels for the PSD net's 32-channel layers live in conv_mfma.hip.
#include "wfs_common.h"

namespace {

constexpr int TB = 256;
constexpr int CT = 8;        // output channels per thread (generic kernel)

struct KMap {                // table column used for filter offset k (identity unless remapped)
    int v[128];
};

// R = capacity (strides, grid); the number of valid rows comes from device memory when r_dev is given
__device__ __forceinline__ long long valid_rows(long long R, const long long *r_dev) {
    long long v = r_dev ? *r_dev : R;
    return v < R ? v : R;
}

// ------------------------------------------------------------------------------------------
// generic gather conv: block = 64 rows x 4 waves; wave w owns output channels
// [cbase + w*CT, cbase + w*CT + CT) of the 32-channel slice blockIdx.y.  The filter values a wave
// needs are wave-uniform -> scalar loads; X values are per-lane row gathers (L1/L2 served).
template <typename T, bool TRANSPOSE_W>
__global__ void __launch_bounds__(TB) k_gather_conv(const int *__restrict__ table, KMap kmap, int K, int identity_k,
                                                    long long R, const long long *__restrict__ r_dev,
                                                    const T *__restrict__ X, int Cx,
                                                    const float *__restrict__ W, int Cw_in, int Cw_out,
                                                    const float *__restrict__ bias, T *__restrict__ Y, int Cy) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long r = (long long)blockIdx.x * 64 + lane;
    const int c0 = blockIdx.y * (4 * CT) + wid * CT;
    if (c0 >= Cy) return;
    const bool live = r < valid_rows(R, r_dev);
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = (bias != nullptr && c0 + c < Cy) ? bias[c0 + c] : 0.f;
    for (int k = 0; k < K; ++k) {
        int nb = -1;
        if (live) nb = (k == identity_k) ? (int)r : table[(long long)kmap.v[k] * R + r];
        if (__ballot(nb >= 0) == 0ull) continue;      // wave-uniform skip of empty offsets
        const float *Wk = W + (long long)k * Cw_in * Cw_out;
        const T *xrow = X + (long long)(nb >= 0 ? nb : 0) * Cx;
        for (int ci = 0; ci < Cx; ++ci) {
            float x = nb >= 0 ? wfs_ld(xrow + ci) : 0.f;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                int co = c0 + c;
                if (co < Cy) {
                    float w = TRANSPOSE_W ? Wk[(long long)co * Cw_out + ci] : Wk[(long long)ci * Cw_out + co];
                    acc[c] = fmaf(x, w, acc[c]);
                }
            }
        }
    }
    if (live) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
            if (c0 + c < Cy) wfs_st(Y + r * Cy + c0 + c, acc[c]);
    }
}

// ------------------------------------------------------------------------------------------
// scatter form (float atomics), only for inputs with duplicate coordinates, where the inverse of
// the gather table is not a function:   Y[table[k][r]] += X[r] . W[k]   (Y fp32, pre-zeroed/biased)
template <typename T, bool TRANSPOSE_W>
__global__ void __launch_bounds__(TB) k_scatter_conv(const int *__restrict__ table, int K, int identity_k,
                                                     long long R, const T *__restrict__ X, int Cx,
                                                     const float *__restrict__ W, int Cw_in, int Cw_out,
                                                     float *__restrict__ Y, int Cy) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long r = (long long)blockIdx.x * 64 + lane;
    const int c0 = blockIdx.y * (4 * CT) + wid * CT;
    if (c0 >= Cy || r >= R) return;
    const T *xrow = X + r * Cx;
    for (int k = 0; k < K; ++k) {
        int dst = (k == identity_k) ? (int)r : table[(long long)k * R + r];
        if (dst < 0) continue;
        const float *Wk = W + (long long)k * Cw_in * Cw_out;
        float acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = 0.f;
        for (int ci = 0; ci < Cx; ++ci) {
            float x = wfs_ld(xrow + ci);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                int co = c0 + c;
                if (co < Cy) {
                    float w = TRANSPOSE_W ? Wk[(long long)co * Cw_out + ci] : Wk[(long long)ci * Cw_out + co];
                    acc[c] = fmaf(x, w, acc[c]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CT; ++c)
            if (c0 + c < Cy) atomicAdd(Y + (long long)dst * Cy + c0 + c, acc[c]);
    }
}

// ------------------------------------------------------------------------------------------
// dW:  part[chunk][k][a][b] = sum_{r in chunk} S[r][a] * G[table[k][r]][b]  over a 32x32 (a,b) tile,
// then a second kernel sums the chunks in fixed order (deterministic).
constexpr int DW_TA = 32, DW_TBB = 32, DW_ROWS = 64;

template <typename T>
__global__ void __launch_bounds__(TB) k_gather_dw(const int *__restrict__ table, int K, int identity_k, long long R,
                                                  const long long *__restrict__ r_dev, long long rows_per_chunk,
                                                  const T *__restrict__ S, int Cs, const T *__restrict__ G, int Cg,
                                                  float *__restrict__ part, int tiles_a, int tiles_b) {
    __shared__ float sS[DW_ROWS][DW_TA + 1];
    __shared__ float sG[DW_ROWS][DW_TBB + 1];
    __shared__ int sNb[DW_ROWS];
    const int k = blockIdx.y;
    const int ta = blockIdx.z / tiles_b, tb = blockIdx.z % tiles_b;
    const int a0 = ta * DW_TA, b0 = tb * DW_TBB;
    const long long chunk = blockIdx.x;
    const long long Rv = valid_rows(R, r_dev);
    const long long r_begin = chunk * rows_per_chunk;
    const long long r_end = r_begin + rows_per_chunk < Rv ? r_begin + rows_per_chunk : Rv;
    // thread -> 2x2 micro tile of the 32x32 output tile
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    for (long long base = r_begin; base < r_end; base += DW_ROWS) {
        int nb = -1;
        if (threadIdx.x < DW_ROWS) {
            long long r = base + threadIdx.x;
            if (r < r_end) nb = (k == identity_k) ? (int)r : table[(long long)k * R + r];
            sNb[threadIdx.x] = nb;
        }
        if (!__syncthreads_or(nb >= 0)) continue;      // block-uniform skip of empty row groups
        for (int e = threadIdx.x; e < DW_ROWS * DW_TA; e += TB) {
            int rr = e / DW_TA, a = e % DW_TA;
            long long r = base + rr;
            float v = 0.f;
            if (sNb[rr] >= 0 && a0 + a < Cs) v = wfs_ld(S + r * Cs + a0 + a);
            sS[rr][a] = v;
        }
        for (int e = threadIdx.x; e < DW_ROWS * DW_TBB; e += TB) {
            int rr = e / DW_TBB, b = e % DW_TBB;
            int src = sNb[rr];
            float v = 0.f;
            if (src >= 0 && b0 + b < Cg) v = wfs_ld(G + (long long)src * Cg + b0 + b);
            sG[rr][b] = v;
        }
        __syncthreads();
#pragma unroll 8
        for (int rr = 0; rr < DW_ROWS; ++rr) {
            float s0 = sS[rr][ty * 2], s1 = sS[rr][ty * 2 + 1];
            float g0 = sG[rr][tx * 2], g1 = sG[rr][tx * 2 + 1];
            acc[0][0] = fmaf(s0, g0, acc[0][0]);
            acc[0][1] = fmaf(s0, g1, acc[0][1]);
            acc[1][0] = fmaf(s1, g0, acc[1][0]);
            acc[1][1] = fmaf(s1, g1, acc[1][1]);
        }
        __syncthreads();
    }
    // part layout: [chunk][k][Cs][Cg]
    float *p = part + ((long long)chunk * K + k) * Cs * Cg;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int a = a0 + ty * 2 + i, b = b0 + tx * 2 + j;
            if (a < Cs && b < Cg) p[(long long)a * Cg + b] = acc[i][j];
        }
}

// dW[k][a][b] (swap==0) or dW[k][b][a] (swap==1) = sum_chunk part[chunk][k][a][b]
__global__ void k_dw_reduce(const float *__restrict__ part, long long nchunks, int K, int Cs, int Cg, int swap,
                            float *__restrict__ dW) {
    long long e = (long long)blockIdx.x * TB + threadIdx.x;
    long long per = (long long)K * Cs * Cg;
    if (e >= per) return;
    float s = 0.f;
    for (long long c = 0; c < nchunks; ++c) s += part[c * per + e];
    int k = (int)(e / ((long long)Cs * Cg));
    int rem = (int)(e % ((long long)Cs * Cg));
    int a = rem / Cg, b = rem % Cg;
    if (swap)
        dW[((long long)k * Cg + b) * Cs + a] = s;
    else
        dW[e] = s;
}


// ------------------------------------------------------------------------------------------ shape-generic MFMA kernels
// Any Cin / Cout, any K, fp32 / bf16 / fp16 rows, exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32: bitwise an fp32 fma
// chain, MI355X_MICROARCH.md "Matrix cores").  These carry the layers the 32-channel kernels of conv_mfma.hip do not:
// the reference's wide 2-D stacks (config/examples/GEP.json: 252 -> 158 -> 64 channels through
// src/models/SPConvBlocks.py:450-516; the SparseConv2DPreserve stacks of 130 ... 154 channels, :730-822) and any
// 16 / 24 / 64-channel 3-D layer.  Channel counts need not be multiples of anything (edge tiles are masked).
typedef float gm_f32x16 __attribute__((ext_vector_type(16)));
constexpr int GM_WAVES = 8;         // waves of a block = shares of the contraction of ONE 32 x 32 output tile
constexpr int GM_CHUNK = 256;       // input channels staged per pass
constexpr int GM_BATCH = 16;        // MFMA steps of a wave per pass: 256 channels = 128 pairs over 8 waves
static_assert(GM_BATCH * GM_WAVES * 2 >= GM_CHUNK, "one batch must cover a wave's share of a pass");

// forward / dX: block = (32 output rows, 32 output channels).  Per active kernel offset the 32 gathered input rows are
// staged once into LDS (whole rows, coalesced), the block's 8 waves take interleaved pairs of the Cin contraction
// (A from LDS, B = filter values straight from L2: 128-B coalesced across a wave for the forward, a lane's own filter
// row for dX), and their 8 accumulators are added through LDS in wave order (deterministic, no atomics, no workspace).
template <typename T, bool TRANSPOSE_W>
__global__ void __launch_bounds__(64 * GM_WAVES) k_gconv_mfma(const int *__restrict__ table, KMap kmap, int K, int identity_k,
                                                              long long R, const long long *__restrict__ r_dev,
                                                              const T *__restrict__ X, int Cx, const float *__restrict__ W,
                                                              int Cw_in, int Cw_out, const float *__restrict__ bias,
                                                              T *__restrict__ Y, int Cy) {
    __shared__ float sA[32][GM_CHUNK + 1];
    __shared__ float sRed[GM_WAVES][1024];
    __shared__ int sNb[128][32];            // the tile's table entries for every offset: ONE round trip, not K
    __shared__ int sAct[128];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long long tile = blockIdx.x;
    const int col0 = blockIdx.y * 32, col = col0 + r;
    const bool col_ok = col < Cy;
    const long long Rv = valid_rows(R, r_dev);
    for (int e = threadIdx.x; e < K * 32; e += 64 * GM_WAVES) {
        const int k = e >> 5;
        const long long row = tile * 32 + (e & 31);
        int nb = -1;
        if (row < Rv) nb = (k == identity_k) ? (int)row : table[(long long)kmap.v[k] * R + row];
        sNb[k][e & 31] = nb;
    }
    __syncthreads();
    for (int k = wid; k < K; k += GM_WAVES) {
        const unsigned long long any = __ballot(lane < 32 && sNb[k][r] >= 0);
        if (lane == 0) sAct[k] = any != 0ull;
    }
    __syncthreads();
    gm_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // A stage = (active offset k, pass c0 over the input channels).  A wave stages rows wid, wid + 8, wid + 16, wid + 24,
    // a lane the columns lane, lane + 64, ... of the pass.  Software pipeline: while the MFMAs of stage s run, the rows of
    // stage s + 1 are already on their way from L2 into registers.
    auto next_stage = [&](int &k, int &c0) {
        c0 += GM_CHUNK;
        if (c0 >= Cx) {
            c0 = 0;
            do ++k; while (k < K && !sAct[k]);
        }
    };
    // fetch() only ASKS for the values (raw, from clamped addresses); validity is applied when they are written to LDS
    // one stage later -- a select right behind the load would make the wave wait for the prefetch before its MFMAs
    float pre[4][GM_CHUNK / 64];
    auto fetch = [&](int k, int c0) {
        const int cn = Cx - c0 < GM_CHUNK ? Cx - c0 : GM_CHUNK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int src = sNb[k][wid + q * GM_WAVES];
            const T *xrow = X + (long long)(src >= 0 ? src : 0) * Cx + c0;
#pragma unroll
            for (int j = 0; j < GM_CHUNK / 64; ++j) {
                const int c = lane + 64 * j;
                pre[q][j] = wfs_ld(xrow + (c < cn ? c : 0));
            }
        }
    };
    int k = -1, c0 = Cx;               // "before the first stage"
    next_stage(k, c0);
    if (k < K) fetch(k, c0);
    while (k < K) {
        const int cn = Cx - c0 < GM_CHUNK ? Cx - c0 : GM_CHUNK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool row_ok = sNb[k][wid + q * GM_WAVES] >= 0;
#pragma unroll
            for (int j = 0; j < GM_CHUNK / 64; ++j)
                sA[wid + q * GM_WAVES][lane + 64 * j] = (row_ok && lane + 64 * j < cn) ? pre[q][j] : 0.f;
        }
        __syncthreads();
        const float *Wk = W + (long long)k * Cw_in * Cw_out;
        const int cc0 = c0;
        // This wave's share of the pass: `iters` <= GM_BATCH consecutive pairs of channels (a pass is at most 256
        // channels = 128 pairs over 8 waves).  Order matters: loads return in order, so the filter values of THIS stage
        // are asked for first, the rows of the NEXT stage second -- the MFMAs then wait for the former only.
        const int iters = (((cn + 1) >> 1) + GM_WAVES - 1) / GM_WAVES;
        float av[GM_BATCH], bv[GM_BATCH];
#pragma unroll
        for (int u = 0; u < GM_BATCH; ++u) {
            const int c = 2 * (wid * iters + u) + h;
            const bool ok = c < cn && u < iters;
            const int cc = ok ? c : 0;
            const long long cw = cc0 + cc;
            const float b = TRANSPOSE_W ? Wk[(long long)(col_ok ? col : 0) * Cw_out + cw]
                                        : Wk[cw * Cw_out + (col_ok ? col : 0)];
            bv[u] = (ok && col_ok) ? b : 0.f;
            av[u] = ok ? sA[r][cc] : 0.f;
        }
        next_stage(k, c0);
        if (k < K) fetch(k, c0);       // in flight during the MFMAs below
#pragma unroll
        for (int u = 0; u < GM_BATCH; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        __syncthreads();
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 16; ++i) sRed[wid][((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += 64 * GM_WAVES) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < GM_WAVES; ++w) v += sRed[w][e];
        const long long row = tile * 32 + (e >> 5);
        const int cc = col0 + (e & 31);
        if (row < Rv && cc < Cy) wfs_st(Y + row * Cy + cc, v + (bias ? bias[cc] : 0.f));
    }
}

// dW: one wave = one 32 x 32 tile of dW[k] over a chunk of rows; the rows are the MFMA contraction (two per
// instruction): A[a][kk] = S[row][a0 + a], B[kk][b] = G[table[k][row]][b0 + b], both whole 128-B row pieces per
// half-wave straight from L2.  Same slab layout as k_gather_dw (part[chunk][k][Cs][Cg]) -> k_dw_reduce.
template <typename T>
__global__ void __launch_bounds__(64) k_gdw_mfma(const int *__restrict__ table, int K, int identity_k, long long R,
                                                 const long long *__restrict__ r_dev, long long rows_per_chunk,
                                                 const T *__restrict__ S, int Cs, const T *__restrict__ G, int Cg,
                                                 float *__restrict__ part, int tiles_b) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int k = blockIdx.y;
    const int a0 = (blockIdx.z / tiles_b) * 32, b0 = (blockIdx.z % tiles_b) * 32;
    const long long chunk = blockIdx.x;
    const long long Rv = valid_rows(R, r_dev);
    const long long r_begin = chunk * rows_per_chunk;
    const long long r_end = r_begin + rows_per_chunk < Rv ? r_begin + rows_per_chunk : Rv;
    const bool a_ok = a0 + j < Cs, b_ok = b0 + j < Cg;
    const int ac = a_ok ? a0 + j : 0, bc = b_ok ? b0 + j : 0;
    gm_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // groups of 64 rows, software-pipelined: the 64 row pieces of group g + 1 are asked for before the 32 MFMAs of group
    // g run (unconditional clamped loads + select: a load under a per-element branch costs a round trip each)
    float av[32], bv[32], an[32], bn[32];
    auto fetch = [&](long long base, float (&A)[32], float (&B)[32]) {
        const long long row = base + lane;
        int nbv = -1;
        if (row < r_end) nbv = (k == identity_k) ? (int)row : table[(long long)k * R + row];
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int n0 = __builtin_amdgcn_readlane(nbv, 2 * s), n1 = __builtin_amdgcn_readlane(nbv, 2 * s + 1);
            const int nb = h ? n1 : n0;
            const long long rr = base + 2 * s + h;
            const float a = wfs_ld(S + (nb >= 0 ? rr : r_begin) * Cs + ac);
            const float b = wfs_ld(G + (long long)(nb >= 0 ? nb : 0) * Cg + bc);
            A[s] = (nb >= 0 && a_ok) ? a : 0.f;
            B[s] = (nb >= 0 && b_ok) ? b : 0.f;
        }
    };
    if (r_begin < r_end) fetch(r_begin, av, bv);
    for (long long base = r_begin; base < r_end; base += 64) {
        const bool more = base + 64 < r_end;
        if (more) fetch(base + 64, an, bn);
#pragma unroll
        for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
        if (more) {
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                av[s] = an[s];
                bv[s] = bn[s];
            }
        }
    }
    float *p = part + ((long long)chunk * K + k) * Cs * Cg;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int a = a0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (a < Cs && b_ok) p[(long long)a * Cg + b0 + j] = acc[i];
    }
}

// shapes the generic MFMA kernels take over from the VALU ones
inline bool gm_shape(int Ca, int Cb) { return Ca >= 8 && Cb >= 8 && (long long)Ca * Cb >= 256; }

long long dw_chunks(long long R) {
    long long chunks = wfs_cdiv(R, 4096);
    if (chunks < 1) chunks = 1;
    if (chunks > 512) chunks = 512;
    return chunks;
}
// the MFMA dW: one wave per (chunk, offset, 32 x 32 tile) -- short inputs get chunks of 256 rows so that the launch has
// enough waves, the slabs stay under 32 MiB
long long dw_chunks_mfma(long long R, int K, int Cs, int Cg) {
    long long chunks = wfs_cdiv(R, 256);
    const long long per = (long long)K * Cs * Cg * 4;
    const long long cap = per > 0 ? (32ll << 20) / per : 1;
    if (chunks > 64) chunks = 64;
    if (chunks > cap) chunks = cap;
    const long long base = dw_chunks(R);
    return chunks < base ? base : chunks;
}

}  // namespace

static int gather_conv_impl(const int32_t *table, const int32_t *kmap_host, int32_t K, int32_t identity_k,
                            int64_t R, const void *X, int64_t X_rows, int32_t Cx, const float *W, int32_t Cw_in,
                            int32_t Cw_out, int32_t transpose_w, const float *bias, void *Y, int32_t dtype,
                            const int64_t *r_dev_, void *stream_, int packed_kl = 0) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(packed_kl == 0 || wfs_gather_packed_ok(packed_kl, K, Cx, transpose_w ? Cw_in : Cw_out, dtype, transpose_w ? 1 : 2),
                WFS_EINVAL, "a packed table (kl %d) is not taken by this product (wfs_gather_packed_ok): expand it with "
                "wfs_unpack_table", packed_kl);
    const long long *r_dev = (const long long *)r_dev_;
    (void)X_rows;
    WFS_REQUIRE(K >= 1 && K <= 128, WFS_EINVAL, "kernel volume %d not in [1,128]", K);
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    const int Cy = transpose_w ? Cw_in : Cw_out;
    WFS_REQUIRE(Cx == (transpose_w ? Cw_out : Cw_in), WFS_EINVAL, "channel mismatch: X has %d, filter wants %d", Cx,
                transpose_w ? Cw_out : Cw_in);
    if (R == 0) return WFS_OK;
    WFS_REQUIRE((table || (K == 1 && identity_k == 0)) && X && W && Y, WFS_EINVAL, "NULL device pointer");
    KMap km;
    for (int k = 0; k < K; ++k) {
        km.v[k] = kmap_host ? kmap_host[k] : k;
        WFS_REQUIRE(km.v[k] >= 0 && km.v[k] < K, WFS_EINVAL, "kmap[%d] out of range", k);
    }
    WfsTimerScope timer(WFS_TIMER_GATHER_CONV, stream);
    // the fast kernels know two column maps: identity, and the SubM mirror k -> K-1-k
    bool is_ident = true, is_mirror = true;
    for (int k = 0; k < K; ++k) {
        is_ident = is_ident && km.v[k] == k;
        is_mirror = is_mirror && km.v[k] == K - 1 - k;
    }
    // (the fp32 kernel addresses the gathered rows through 32-bit byte offsets: fewer than 2^24 rows of 128 B)
    if (dtype == WFS_F32 && Cx == 32 && Cy == 32 && wfs_mfma_gconv32_ok(K) && table && (is_ident || is_mirror) &&
        X_rows < (1ll << 24)) {
        return wfs_launch_gconv32_f32(table, is_ident ? 0 : 1, K, identity_k, R, r_dev, (const float *)X, W, transpose_w,
                                      bias, (float *)Y, stream, packed_kl);
    }
    if (dtype != WFS_F32 && Cx == 32 && Cy == 32 && K <= 27 && table && (is_ident || is_mirror) && X_rows < (1ll << 25)) {
        return wfs_launch_gconv32_h16(table, is_ident ? 0 : 1, K, identity_k, R, r_dev, X, W, transpose_w, bias, Y,
                                      dtype, stream, packed_kl);
    }
    WFS_REQUIRE(packed_kl == 0, WFS_EINVAL, "a packed table reached a kernel that reads dense ones");
    if (Cx == 2 && Cy == 32 && !transpose_w && table)
        return wfs_launch_gconv_c2c32(table, kmap_host, K, identity_k, R, r_dev, X, W, bias, Y, dtype, stream);
    if (gm_shape(Cx, Cy) && table) {
        const dim3 g((unsigned)wfs_cdiv(R, 32), (unsigned)wfs_cdiv(Cy, 32)), b(64 * GM_WAVES);
#define WFS_GM(T, TR)                                                                                            \
    k_gconv_mfma<T, TR><<<g, b, 0, stream>>>(table, km, K, identity_k, R, r_dev, (const T *)X, Cx, W, Cw_in, Cw_out, \
                                             bias, (T *)Y, Cy)
        if (dtype == WFS_F32) {
            if (transpose_w) WFS_GM(float, true); else WFS_GM(float, false);
        } else if (dtype == WFS_BF16) {
            if (transpose_w) WFS_GM(wfs_bf16, true); else WFS_GM(wfs_bf16, false);
        } else {
            if (transpose_w) WFS_GM(wfs_f16, true); else WFS_GM(wfs_f16, false);
        }
#undef WFS_GM
        WFS_LAUNCH_CHECK();
        return WFS_OK;
    }
    dim3 grid((unsigned)wfs_cdiv(R, 64), (unsigned)wfs_cdiv(Cy, 4 * CT)), block(TB);
#define WFS_GC(T, TR)                                                                                           \
    k_gather_conv<T, TR><<<grid, block, 0, stream>>>(table, km, K, identity_k, R, r_dev, (const T *)X, Cx, W,  \
                                                      Cw_in, Cw_out, bias, (T *)Y, Cy)
    if (dtype == WFS_F32) {
        if (transpose_w) WFS_GC(float, true); else WFS_GC(float, false);
    } else if (dtype == WFS_BF16) {
        if (transpose_w) WFS_GC(wfs_bf16, true); else WFS_GC(wfs_bf16, false);
    } else {
        if (transpose_w) WFS_GC(wfs_f16, true); else WFS_GC(wfs_f16, false);
    }
#undef WFS_GC
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

extern "C" int wfs_gather_conv(const int32_t *table, const int32_t *kmap_host, int32_t K, int32_t identity_k,
                               int64_t R, const void *X, int64_t X_rows, int32_t Cx, const float *W, int32_t Cw_in,
                               int32_t Cw_out, int32_t transpose_w, const float *bias, void *Y, int32_t dtype,
                               const int64_t *r_dev, int32_t packed_kl, void *stream) {
    WFS_REQUIRE(packed_kl == 0 || !kmap_host, WFS_EINVAL, "a packed table takes no column map");
    return gather_conv_impl(table, kmap_host, K, identity_k, R, X, X_rows, Cx, W, Cw_in, Cw_out, transpose_w, bias, Y,
                            dtype, r_dev, stream, packed_kl);
}

// which = 1: wfs_gather_conv with transpose_w (dX), 2: wfs_gather_conv forward, 3: wfs_gather_dw
extern "C" int wfs_gather_packed_ok(int32_t packed_kl, int32_t K, int32_t Ca, int32_t Cb, int32_t dtype, int32_t which) {
    if (packed_kl < 1 || packed_kl > 8 || K < 1 || K > 32 || K % packed_kl != 0 || Ca != 32 || Cb != 32 || !wfs_dtype_ok(dtype))
        return 0;
    if (which == 3) return 1;
    return which == 1 && packed_kl == 3 && K <= 27;
}

extern "C" int wfs_scatter_conv(const int32_t *table, int32_t K, int32_t identity_k, int64_t R, const void *X,
                                int32_t Cx, const float *W, int32_t Cw_in, int32_t Cw_out, int32_t transpose_w,
                                float *Y_accum, int32_t dtype, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(K >= 1, WFS_EINVAL, "bad K");
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    const int Cy = transpose_w ? Cw_in : Cw_out;
    WFS_REQUIRE(Cx == (transpose_w ? Cw_out : Cw_in), WFS_EINVAL, "channel mismatch");
    if (R == 0) return WFS_OK;
    WFS_REQUIRE(table && X && W && Y_accum, WFS_EINVAL, "NULL device pointer");
    dim3 grid((unsigned)wfs_cdiv(R, 64), (unsigned)wfs_cdiv(Cy, 4 * CT)), block(TB);
#define WFS_SC(T, TR)                                                                                          \
    k_scatter_conv<T, TR><<<grid, block, 0, stream>>>(table, K, identity_k, R, (const T *)X, Cx, W, Cw_in, Cw_out, \
                                                       Y_accum, Cy)
    if (dtype == WFS_F32) {
        if (transpose_w) WFS_SC(float, true); else WFS_SC(float, false);
    } else if (dtype == WFS_BF16) {
        if (transpose_w) WFS_SC(wfs_bf16, true); else WFS_SC(wfs_bf16, false);
    } else {
        if (transpose_w) WFS_SC(wfs_f16, true); else WFS_SC(wfs_f16, false);
    }
#undef WFS_SC
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

extern "C" size_t wfs_gather_dw_workspace_bytes(int32_t K, int64_t R, int32_t Cs, int32_t Cg) {
    size_t generic = (size_t)(gm_shape(Cs, Cg) ? dw_chunks_mfma(R, K, Cs, Cg) : dw_chunks(R)) * K * Cs * Cg * sizeof(float);
    size_t fast = wfs_dw_fast_workspace(K, R, Cs, Cg);
    if (wfs_wide_dw_ok(K, R, Cs, Cg, WFS_F32)) {                  // fp32 rows need the larger staging copies
        size_t wide = wfs_wide_dw_workspace(K, R, Cs, Cg, WFS_F32);
        generic = generic > wide ? generic : wide;
    }
    return generic > fast ? generic : fast;
}

static int gather_dw_impl(const int32_t *table, const int32_t *kmap_host, int32_t K, int32_t identity_k, int64_t R,
                          const void *S, int32_t Cs, const void *G, int64_t G_rows, int32_t Cg, int32_t swap, float *dW,
                          int32_t dtype, void *workspace, size_t workspace_bytes, const int64_t *r_dev_,
                          wfs_dw_job *defer, void *stream_, int packed_kl = 0) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(packed_kl == 0 || (wfs_gather_packed_ok(packed_kl, K, Cs, Cg, dtype, 3) && !kmap_host), WFS_EINVAL,
                "a packed table (kl %d) is not taken by this product (wfs_gather_packed_ok): expand it with wfs_unpack_table",
                packed_kl);
    const long long *r_dev = (const long long *)r_dev_;
    WFS_REQUIRE(K >= 1 && K <= 65535, WFS_EINVAL, "bad K");
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    WFS_REQUIRE(dW, WFS_EINVAL, "NULL dW");
    if (defer) *defer = wfs_dw_job{nullptr, 0, 0, 0, 0, 0, 0, nullptr};        // nothing pending unless a fast path says so
    if (R == 0) {
        WFS_HIP_CHECK(hipMemsetAsync(dW, 0, (size_t)K * Cs * Cg * sizeof(float), stream));
        return WFS_OK;
    }
    WFS_REQUIRE((table || (K == 1 && identity_k == 0)) && S && G && workspace, WFS_EINVAL, "NULL device pointer");
    size_t need = wfs_gather_dw_workspace_bytes(K, R, Cs, Cg);
    WFS_REQUIRE(workspace_bytes >= need, WFS_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, need);
    WfsTimerScope timer(WFS_TIMER_GATHER_DW, stream);
    if (!packed_kl && wfs_wide_dw_ok(K, R, Cs, Cg, dtype) && K <= 128 && ((uintptr_t)workspace & 15) == 0) {
        for (int k = 0; k < K && kmap_host; ++k)
            WFS_REQUIRE(kmap_host[k] >= 0 && kmap_host[k] < K, WFS_EINVAL, "kmap[%d] out of range", k);
        return wfs_launch_wide_dw(table, kmap_host, K, identity_k, R, r_dev, S, Cs, G, G_rows, Cg, swap, dW, dtype, workspace,
                                  workspace_bytes, stream);
    }
    // the fp32 kernel addresses S and G through 32-bit buffer offsets (128 B per row): fewer than 2^24 rows each
    const bool rows_fit = dtype != WFS_F32 || (R < (1ll << 24) && G_rows < (1ll << 24));
    if (Cs == 32 && Cg == 32 && table && !kmap_host && rows_fit)
        return wfs_launch_gdw32(table, K, identity_k, R, r_dev, S, G, swap, dW, (float *)workspace, dtype, defer, stream,
                                packed_kl);
    WFS_REQUIRE(packed_kl == 0, WFS_EINVAL, "a packed table reached a kernel that reads dense ones");
    bool is_ident = true, is_mirror = true;
    for (int k = 0; k < K && kmap_host; ++k) {
        is_ident = is_ident && kmap_host[k] == k;
        is_mirror = is_mirror && kmap_host[k] == K - 1 - k;
    }
    if (!kmap_host) is_mirror = false;
    if (Cs == 32 && Cg == 2 && K <= 27 && table && (is_ident || is_mirror))
        return wfs_launch_gdw_c32c2(table, is_ident ? 0 : 1, K, identity_k, R, r_dev, S, G, swap, dW, (float *)workspace,
                                    dtype, defer, stream);
    WFS_REQUIRE(is_ident, WFS_EINVAL, "a column map is only supported by the 32 x 2 dW kernel");
    long long chunks = gm_shape(Cs, Cg) ? dw_chunks_mfma(R, K, Cs, Cg) : dw_chunks(R);
    long long rows_per_chunk = wfs_cdiv(wfs_cdiv(R, chunks), DW_ROWS) * DW_ROWS;
    int tiles_a = (int)wfs_cdiv(Cs, DW_TA), tiles_b = (int)wfs_cdiv(Cg, DW_TBB);
    WFS_REQUIRE((long long)tiles_a * tiles_b <= 65535, WFS_EINVAL, "channel tile grid too large");
    dim3 grid((unsigned)chunks, (unsigned)K, (unsigned)(tiles_a * tiles_b)), block(TB);
    float *part = (float *)workspace;
    if (gm_shape(Cs, Cg) && table) {
        const dim3 b64(64);
        if (dtype == WFS_F32)
            k_gdw_mfma<float><<<grid, b64, 0, stream>>>(table, K, identity_k, R, r_dev, rows_per_chunk, (const float *)S, Cs,
                                                        (const float *)G, Cg, part, tiles_b);
        else if (dtype == WFS_BF16)
            k_gdw_mfma<wfs_bf16><<<grid, b64, 0, stream>>>(table, K, identity_k, R, r_dev, rows_per_chunk,
                                                           (const wfs_bf16 *)S, Cs, (const wfs_bf16 *)G, Cg, part, tiles_b);
        else
            k_gdw_mfma<wfs_f16><<<grid, b64, 0, stream>>>(table, K, identity_k, R, r_dev, rows_per_chunk, (const wfs_f16 *)S,
                                                          Cs, (const wfs_f16 *)G, Cg, part, tiles_b);
    } else if (dtype == WFS_F32)
        k_gather_dw<float><<<grid, block, 0, stream>>>(table, K, identity_k, R, r_dev, rows_per_chunk, (const float *)S,
                                                       Cs, (const float *)G, Cg, part, tiles_a, tiles_b);
    else if (dtype == WFS_BF16)
        k_gather_dw<wfs_bf16><<<grid, block, 0, stream>>>(table, K, identity_k, R, r_dev, rows_per_chunk,
                                                          (const wfs_bf16 *)S, Cs, (const wfs_bf16 *)G, Cg, part,
                                                          tiles_a, tiles_b);
    else
        k_gather_dw<wfs_f16><<<grid, block, 0, stream>>>(table, K, identity_k, R, r_dev, rows_per_chunk,
                                                         (const wfs_f16 *)S, Cs, (const wfs_f16 *)G, Cg, part, tiles_a,
                                                         tiles_b);
    WFS_LAUNCH_CHECK();
    long long per = (long long)K * Cs * Cg;
    k_dw_reduce<<<dim3((unsigned)wfs_cdiv(per, TB)), block, 0, stream>>>(part, chunks, K, Cs, Cg, swap, dW);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

extern "C" int wfs_gather_dw(const int32_t *table, const int32_t *kmap_host, int32_t K, int32_t identity_k, int64_t R,
                             const void *S, int32_t Cs, const void *G, int64_t G_rows, int32_t Cg, int32_t swap, float *dW,
                             int32_t dtype, void *workspace, size_t workspace_bytes, const int64_t *r_dev,
                             wfs_dw_job *defer, int32_t packed_kl, void *stream) {
    return gather_dw_impl(table, kmap_host, K, identity_k, R, S, Cs, G, G_rows, Cg, swap, dW, dtype, workspace,
                          workspace_bytes, r_dev, defer, stream, packed_kl);
}

// torch.ops.spconv.indice_conv_backward as ONE call: dW (as wfs_gather_dw, swap == 0) and dX (as wfs_gather_conv with
// transpose_w) of a conv / SubM layer, both through the by-input table.  32 -> 32 layers with 16-bit rows run both
// products in one launch (conv_mfma.hip k_bwd32_bf16); every other shape runs the two entry points one after the other.
extern "C" int wfs_conv_backward(const int32_t *table, int32_t K, int32_t identity_k, int64_t R, const void *X,
                                 const void *dY, int64_t dY_rows, int32_t Cin, int32_t Cout, const float *W, void *dX,
                                 float *dW, int32_t dtype, void *workspace, size_t workspace_bytes, const int64_t *r_dev,
                                 wfs_dw_job *defer, int32_t packed_kl, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(dX && dW, WFS_EINVAL, "wfs_conv_backward computes both gradients (use wfs_gather_conv / wfs_gather_dw for one)");
    const long long row_lim = dtype == WFS_F32 ? (1ll << 24) : (1ll << 25);       // 32-bit byte offsets of the gathers
    const bool fused = Cin == 32 && Cout == 32 && table && wfs_bwd32_fused_ok(K, packed_kl, dtype) && R > 0 &&
                       R < row_lim && dY_rows < row_lim && (packed_kl == 0 || identity_k < 0);
    if (!fused) {
        int rc = gather_dw_impl(table, nullptr, K, identity_k, R, X, Cin, dY, dY_rows, Cout, 0, dW, dtype, workspace,
                                workspace_bytes, r_dev, defer, stream_, packed_kl);
        if (rc != WFS_OK) return rc;
        return gather_conv_impl(table, nullptr, K, identity_k, R, dY, dY_rows, Cout, W, Cin, Cout, 1, nullptr, dX, dtype, r_dev,
                                stream_, packed_kl);
    }
    WFS_REQUIRE(X && dY && W && workspace, WFS_EINVAL, "NULL device pointer");
    const size_t need = wfs_gather_dw_workspace_bytes(K, R, Cin, Cout);
    WFS_REQUIRE(workspace_bytes >= need, WFS_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, need);
    if (defer) *defer = wfs_dw_job{nullptr, 0, 0, 0, 0, 0, 0, nullptr};
    WfsTimerScope timer(WFS_TIMER_CONV_BACKWARD, stream);
    return wfs_launch_bwd32_h16(table, packed_kl, K, identity_k, R, (const long long *)r_dev, X, dY, W, dX, 0, dW,
                                (float *)workspace, dtype, defer, stream);
}

extern "C" int wfs_dw_reduce_jobs(const wfs_dw_job *jobs, int32_t n, void *stream) {
    WFS_REQUIRE(n >= 0 && n <= 16 && (n == 0 || jobs), WFS_EINVAL, "%d jobs (0 .. 16)", n);
    wfs_dw_job live[16];
    int m = 0;
    for (int i = 0; i < n; ++i) {
        if (jobs[i].nslabs <= 0) continue;
        WFS_REQUIRE(jobs[i].part && jobs[i].dW && jobs[i].per > 0 && jobs[i].A > 0 && jobs[i].B > 0, WFS_EINVAL,
                    "job %d is incomplete", i);
        live[m++] = jobs[i];
    }
    WfsTimerScope timer(WFS_TIMER_GATHER_DW, (hipStream_t)stream);
    return wfs_launch_dw_jobs(live, m, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------ conv bias gradient
// db[c] = sum over the valid rows of dY[r][c] (spconv's backward leaves this to autograd's `+ bias`; reference layers
// built with trainable_weights=True have a bias, SPConvBlocks.py:498).  Two launches, deterministic: per-block column
// sums over interleaved rows, then one block adds the partials in block order.  workspace: 128 * C floats.
namespace {
constexpr int CS_BLOCKS = 128;
template <typename T>
__global__ void __launch_bounds__(256) k_colsum_partial(const T *__restrict__ X, long long Rcap,
                                                        const long long *__restrict__ r_dev, int C,
                                                        float *__restrict__ partial) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    const long long R = r_dev ? (*r_dev < Rcap ? *r_dev : Rcap) : Rcap;
    if (c >= C) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;              // four independent chains, added in a fixed order
    long long r = blockIdx.x;
    const long long st = gridDim.x;
    for (; r + 3 * st < R; r += 4 * st) {
        s0 += wfs_ld(X + r * C + c);
        s1 += wfs_ld(X + (r + st) * C + c);
        s2 += wfs_ld(X + (r + 2 * st) * C + c);
        s3 += wfs_ld(X + (r + 3 * st) * C + c);
    }
    for (; r < R; r += st) s0 += wfs_ld(X + r * C + c);
    partial[(long long)blockIdx.x * C + c] = (s0 + s1) + (s2 + s3);
}
__global__ void __launch_bounds__(256) k_colsum_fold(const float *__restrict__ partial, int nblk, int C,
                                                     float *__restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += partial[(long long)b * C + c];
    out[c] = s;
}
}  // namespace

extern "C" size_t wfs_column_sum_workspace_bytes(int32_t C) { return (size_t)CS_BLOCKS * (size_t)(C > 0 ? C : 1) * sizeof(float); }

extern "C" int wfs_column_sum(const void *X, int64_t R, int32_t C, float *out, void *workspace, size_t workspace_bytes,
                              int32_t dtype, const int64_t *r_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    WFS_REQUIRE(C >= 1 && out, WFS_EINVAL, "bad channel count %d / NULL output", C);
    if (R == 0) {
        WFS_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)C * sizeof(float), stream));
        return WFS_OK;
    }
    WFS_REQUIRE(X && workspace, WFS_EINVAL, "NULL device pointer");
    WFS_REQUIRE(workspace_bytes >= wfs_column_sum_workspace_bytes(C), WFS_EWORKSPACE, "workspace too small");
    long long nblk = wfs_cdiv(R, 64);
    if (nblk > CS_BLOCKS) nblk = CS_BLOCKS;
    const dim3 grid((unsigned)nblk, (unsigned)wfs_cdiv(C, 256)), block(256);
    float *partial = (float *)workspace;
    const long long *rd = (const long long *)r_dev;
    if (dtype == WFS_F32)
        k_colsum_partial<float><<<grid, block, 0, stream>>>((const float *)X, R, rd, C, partial);
    else if (dtype == WFS_BF16)
        k_colsum_partial<wfs_bf16><<<grid, block, 0, stream>>>((const wfs_bf16 *)X, R, rd, C, partial);
    else
        k_colsum_partial<wfs_f16><<<grid, block, 0, stream>>>((const wfs_f16 *)X, R, rd, C, partial);
    k_colsum_fold<<<dim3((unsigned)wfs_cdiv(C, 256)), block, 0, stream>>>(partial, (int)nblk, C, out);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}
