// wide.hip -- sparse convolutions with hundreds to thousands of channels on the 16-bit matrix cores.
//
// The reference's hybrid net (BASELINE configs[4]: TemporalConvNet front end -> SparseConv2d 2048 -> 1697 (1 x 1)
// -> 1021 -> 345 (3 x 3), src/models/SPConvBlocks.py:450-516 through spconv's indice_conv) spends its time in
// products of tens of GFLOP per layer.  A 32-row x 32-column block (gather_conv.hip) re-reads 62 MB of filters per
// row tile there; this file carries those layers instead:
//
//   k_gemm16      C[z] = A[z] . B[z]^T on v_mfma_f32_32x32x16_{bf16,f16}: 128 x 128 output tile, 64-deep steps, 4 waves
//                 (64 x 64 each), double-buffered LDS filled from a register stage (next tile's global loads in flight
//                 under the MFMAs of this one), fp32 accumulate.  Either operand may be stored contraction-contiguous
//                 ([rows][k], read back with ds_read_b128 from an XOR-swizzled 128-B-row image) or contraction-major
//                 ([k][rows], read back TRANSPOSED with ds_read_b64_tr_b16 from a 256-B-row image): forward, dX and dW
//                 all take the filters and the rows in the layout they already have -- no transposed copies.
//   k_pad_rows    16-bit rows of any channel count (2-byte aligned) -> 16-byte aligned, zero-padded rows, gathered
//                 through a table when given (aligned dword loads + funnel shift, no 2-byte loads)
//   k_pad_f32     fp32 filters -> zero-padded 16-bit rows
//   k_sum_rows    the ordered sum over kernel offsets (fp32), bias, conversion to the row type
//
// A layer is ONE dense product over the side with FEWER rows (R_s source rows, R_d destination rows):
//   destination side shorter (R_d <= R_s):  G[r, (k, c)] = X[table[k][r], c]          (k_pad_rows, 16-bit)
//                                           Y = G . Wcat                               (k_gemm16, offsets = K-segments)
//   source side shorter      (R_s <  R_d):  T[s, (k, c')] = X[s, :] . W[k]             (k_gemm16, offsets = batches)
//                                           Y[r, c'] = bias + sum_k T[table[k][r], (k, c')]   (k_sum_rows, fixed order)
// Both read only the destination-indexed table the gather kernels use (nbr_in for the forward pass, nbr_out for dX),
// write every output row once and use no atomics: results are run-to-run identical.
// dW[k] = S^T . G_k contracts over the rows: both operands contraction-major, one batch per offset, written straight
// into the gradient.
#include "wfs_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int GT = 128;           // output tile edge
constexpr int GK = 64;            // contraction depth of one step
constexpr int GTHREADS = 256;     // 4 waves: 2 (rows) x 2 (columns) of 64 x 64
constexpr int G_TILE_BYTES = GT * GK * 2;            // one operand tile in LDS: 16 KiB
constexpr int G_LDS_BYTES = 4 * G_TILE_BYTES;        // 2 operands x 2 buffers

struct GemmArgs {
    const unsigned short *A, *B;
    void *C;
    const float *bias;            // [N] or NULL (16-bit output only)
    long long lda, ldb, ldc;      // row pitches in elements (lda, ldb: multiples of 8)
    long long sA, sB;             // element offset of segment s: A + s * sA, B + s * sB
    long long zC;                 // element offset of batch z in C
    int M, N, Ks;                 // Ks: contraction length of ONE segment
    int nseg_total, nseg;         // segments in all / per batch: batch z takes segments z * nseg ...
    int nz;
    const long long *k_dev;       // optional device-side contraction length (<= Ks): rows of a dW product
    int out_h;                    // 1: C holds 16-bit elements (bias added), 0: fp32
    int accumulate;               // fp32 output: C += product
};

template <typename H>
__device__ __forceinline__ f32x16 mfma16(s16x8 a, s16x8 b, f32x16 acc) {
    if constexpr (__is_same(H, wfs_f16))
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0,
                                                      0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc,
                                                       0, 0, 0);
}

__device__ __forceinline__ uint4 keep_if(uint4 v, bool ok) {      // component-wise: a vector select goes through scratch
    v.x = ok ? v.x : 0u;
    v.y = ok ? v.y : 0u;
    v.z = ok ? v.z : 0u;
    v.w = ok ? v.w : 0u;
    return v;
}

// One operand tile (128 rows of the output dimension x 64 of the contraction) from global memory into 4 x 16 bytes per
// thread.  KM == false: stored [row][k]: a thread takes piece (t & 7) of rows (t >> 3) + 32 i -- 8 lanes read one full
// 128-B line.  KM == true: stored [k][row]: chunk (t & 15) of contraction rows (t >> 4) + 16 i -- 16 lanes read 256 B.
// Rows / columns beyond the matrix edge are read from a clamped address (their products are never stored); contraction
// indices beyond the segment give zeros.
template <bool KM>
__device__ __forceinline__ void g_load(uint4 (&r)[4], const unsigned short *__restrict__ base, long long ld, int o0,
                                       int lim, int kk0, int Ks, int t) {
    if constexpr (!KM) {
        const int p = t & 7, rr = t >> 3;
        const bool kok = kk0 + p * 8 < Ks;
        const int kk = kok ? kk0 + p * 8 : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = o0 + rr + 32 * i;
            row = row < lim ? row : lim - 1;
            const uint4 v = *reinterpret_cast<const uint4 *>(base + (long long)row * ld + kk);
            r[i] = keep_if(v, kok);
        }
    } else {
        const int ch = t & 15, rr = t >> 4;
        int col = o0 + ch * 8;
        col = col < lim ? col : o0;                  // whole chunks beyond the edge: any valid address
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = kk0 + rr + 16 * i;
            const bool ok = kk < Ks;
            const uint4 v = *reinterpret_cast<const uint4 *>(base + (long long)(ok ? kk : 0) * ld + col);
            r[i] = keep_if(v, ok);
        }
    }
}

// LDS images (byte offsets inside one 16-KiB tile):
//   [row][k]: 128-B rows, 16-B piece p of row r at r * 128 + ((p ^ ((r >> 1) & 7)) << 4): the 16-lane groups of
//             ds_read_b128 ({0-3,12-15,20-27}, ...; MI355X_MICROARCH.md "LDS") land on 16 distinct 16-B bank groups
//   [k][row]: 256-B rows, 16-B chunk c of contraction row k at k * 256 + ((c ^ ((k & 3) << 2)) << 4): the 4 rows x 32
//             columns one half-wave takes with ds_read_b64_tr_b16 cover all 64 banks once
template <bool KM>
__device__ __forceinline__ void s_store(unsigned char *tile, const uint4 (&r)[4], int t) {
    if constexpr (!KM) {
        const int p = t & 7, rr = t >> 3;
        unsigned char *d = tile + rr * 128 + ((p ^ ((rr >> 1) & 7)) << 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4 *>(d + i * 32 * 128) = r[i];
    } else {
        const int ch = t & 15, rr = t >> 4;
        unsigned char *d = tile + rr * 256 + ((ch ^ ((rr & 3) << 2)) << 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4 *>(d + i * 16 * 256) = r[i];
    }
}

// MFMA operand (32 rows of the output dimension starting at o, contraction sub-step s of 16) of lane (i = lane & 31,
// h = lane >> 5): elements k = 16 s + 8 h .. + 7 of row o + i.
template <bool KM>
__device__ __forceinline__ s16x8 frag(const unsigned char *tile, int o, int s, int lane) {
    if constexpr (!KM) {
        const int row = o + (lane & 31), p = 2 * s + (lane >> 5);
        return *reinterpret_cast<const s16x8 *>(tile + row * 128 + ((p ^ ((row >> 1) & 7)) << 4));
    } else {
        // 16-lane group g takes the 4 x 16 block (contraction rows kb .. kb + 3, columns o + 16 (g & 1) ...): lane
        // 4 q + p of the group supplies the address of row q, columns 4 p .. 4 p + 3 and receives column (lane & 15)
        const int g = lane >> 4, l = lane & 15, q = l >> 2, p = l & 3;
        const int kk = 16 * s + 8 * (g >> 1) + q;
        const int col = o + 16 * (g & 1) + 4 * p;
        const unsigned char *a = tile + kk * 256 + (((col >> 3) ^ ((kk & 3) << 2)) << 4) + 8 * ((col >> 2) & 1);
        typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + 4 * 256));
        return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
}

template <typename H, bool A_KM, bool B_KM>
__global__ void __launch_bounds__(GTHREADS) k_gemm16(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w & 1, wn = w >> 1;
    // consecutive block ids go round the 8 XCDs: give every XCD one contiguous range of tiles, rows fastest, so that
    // the blocks sharing a B panel (the large operand: filters / gathered rows) sit behind one L2
    const int tiles_m = (g.M + GT - 1) / GT, tiles_n = (g.N + GT - 1) / GT;
    const long long nblk = (long long)tiles_m * tiles_n * g.nz;
    const long long per = (nblk + 7) >> 3;
    const long long id = (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (id >= nblk) return;
    const int tm = (int)(id % tiles_m);
    const int tn = (int)((id / tiles_m) % tiles_n);
    const int z = (int)(id / ((long long)tiles_m * tiles_n));
    const int m0 = tm * GT, n0 = tn * GT;
    int Ks = g.Ks;
    if (g.k_dev) {
        const long long kd = *g.k_dev;
        Ks = kd < Ks ? (int)(kd < 0 ? 0 : kd) : Ks;
    }
    const int seg0 = z * g.nseg;
    const int nseg = g.nseg_total - seg0 < g.nseg ? g.nseg_total - seg0 : g.nseg;
    const int per_seg = (Ks + GK - 1) / GK;
    const int nsteps = nseg * per_seg;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    uint4 ra[4], rb[4];
    auto fetch = [&](int st) {
        const int seg = seg0 + st / per_seg, kk0 = (st % per_seg) * GK;
        g_load<A_KM>(ra, g.A + seg * g.sA, g.lda, m0, g.M, kk0, Ks, t);
        g_load<B_KM>(rb, g.B + seg * g.sB, g.ldb, n0, g.N, kk0, Ks, t);
    };
    auto park = [&](int buf) {
        s_store<A_KM>(lds + buf * 2 * G_TILE_BYTES, ra, t);
        s_store<B_KM>(lds + buf * 2 * G_TILE_BYTES + G_TILE_BYTES, rb, t);
    };
    if (nsteps > 0) {
        fetch(0);
        park(0);
    }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const bool more = st + 1 < nsteps;
        if (more) fetch(st + 1);                       // in flight under the MFMAs below
        const unsigned char *ta = lds + (st & 1) * 2 * G_TILE_BYTES, *tb = ta + G_TILE_BYTES;
#pragma unroll
        for (int s = 0; s < GK / 16; ++s) {
            const s16x8 a0 = frag<A_KM>(ta, wm * 64, s, lane), a1 = frag<A_KM>(ta, wm * 64 + 32, s, lane);
            const s16x8 b0 = frag<B_KM>(tb, wn * 64, s, lane), b1 = frag<B_KM>(tb, wn * 64 + 32, s, lane);
            acc[0][0] = mfma16<H>(a0, b0, acc[0][0]);
            acc[0][1] = mfma16<H>(a0, b1, acc[0][1]);
            acc[1][0] = mfma16<H>(a1, b0, acc[1][0]);
            acc[1][1] = mfma16<H>(a1, b1, acc[1][1]);
        }
        if (more) park((st + 1) & 1);                  // the other buffer: its readers passed the last barrier
        __syncthreads();
    }
    // C/D map of the 32 x 32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int h = lane >> 5;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wn * 64 + b * 32 + (lane & 31);
            if (n >= g.N) continue;
            const float bv = (g.out_h && g.bias) ? g.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + wm * 64 + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (m >= g.M) continue;
                const long long e = g.zC * z + (long long)m * g.ldc + n;
                if (g.out_h)
                    wfs_st(reinterpret_cast<H *>(g.C) + e, acc[a][b][i] + bv);
                else if (g.accumulate)
                    reinterpret_cast<float *>(g.C)[e] += acc[a][b][i];
                else
                    reinterpret_cast<float *>(g.C)[e] = acc[a][b][i];
            }
        }
}

// ------------------------------------------------------------------------------------------ staging kernels
struct KMapW {
    int v[128];
};

// dst[r, k * Cp + c] = src[row(k, r), c] for c < C, 0 for C <= c < Cp; row(k, r) = table[kmap[k]][r] (r itself at
// identity_k or without a table; -1 or r beyond the valid count: a zero row).  One thread = one 16-byte piece of dst.
// src rows are only 2-byte aligned (odd channel counts): a piece is read as 5 aligned dwords and funnel-shifted.
__global__ void __launch_bounds__(256) k_pad_rows(const int *__restrict__ table, KMapW kmap, int K, int identity_k,
                                                  long long R, const long long *__restrict__ r_dev,
                                                  const unsigned short *__restrict__ src, long long src_rows, int C,
                                                  int Cp, unsigned short *__restrict__ dst) {
    const int pieces = Cp >> 3;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = R * K * pieces;
    if (e >= total) return;
    const int pc = (int)(e % pieces);
    const int k = (int)((e / pieces) % K);
    const long long r = e / ((long long)pieces * K);
    long long Rv = r_dev ? *r_dev : R;
    Rv = Rv < R ? Rv : R;
    long long s = -1;
    if (r < Rv) s = (!table || k == identity_k) ? r : (long long)table[(long long)kmap.v[k] * R + r];
    uint4 out = {0u, 0u, 0u, 0u};
    if (s >= 0 && s < src_rows) {
        const int c0 = pc * 8;
        // byte addresses: the buffer itself may start on an odd element (a view into a larger tensor)
        const uintptr_t a0 = (uintptr_t)src + 2 * (uintptr_t)(s * C + c0);
        const uintptr_t amax = ((uintptr_t)src + 2 * (uintptr_t)(src_rows * (long long)C - 1)) & ~(uintptr_t)3;
        const uintptr_t d0 = a0 & ~(uintptr_t)3;
        unsigned wv[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const uintptr_t a = d0 + 4 * j;
            wv[j] = *reinterpret_cast<const unsigned *>(a <= amax ? a : amax);
        }
        unsigned o[4];
        if (a0 & 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = __builtin_amdgcn_alignbit(wv[j + 1], wv[j], 16);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = wv[j];
        }
        const int n = C - c0;                                  // valid elements of this piece (may be >= 8)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (2 * j >= n) o[j] = 0u;
            else if (2 * j + 1 >= n) o[j] &= 0xFFFFu;
        }
        out = uint4{o[0], o[1], o[2], o[3]};
    }
    *reinterpret_cast<uint4 *>(dst + (r * K + k) * (long long)Cp + pc * 8) = out;
}

// fp32 [rows, C] -> 16-bit [rows, Cp], zero padded (the filters: rows = K * Cw_in, C = Cw_out)
template <typename H>
__global__ void __launch_bounds__(256) k_pad_f32(const float *__restrict__ src, long long rows, int C, int Cp,
                                                 H *__restrict__ dst) {
    const int pieces = Cp >> 3;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * pieces) return;
    const int pc = (int)(e % pieces);
    const long long r = e / pieces;
    const float *s = src + r * C + pc * 8;
    const int n = C - pc * 8;
    unsigned o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float lo = 2 * j < n ? s[2 * j] : 0.f, hi = 2 * j + 1 < n ? s[2 * j + 1] : 0.f;
        o[j] = wfs_pack2<H>(lo, hi);
    }
    *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned short *>(dst) + r * Cp + pc * 8) = uint4{o[0], o[1], o[2], o[3]};
}

// Y[r, c] = bias[c] + sum_k T[row(k, r) * row_pitch + k * k_pitch + c]  in the fixed order k = 0 .. K - 1 (fp32), stored
// as H.  row(k, r) as in k_pad_rows (a missing neighbour contributes nothing).  One block per output row.
template <typename H>
__global__ void __launch_bounds__(256) k_sum_rows(const int *__restrict__ table, KMapW kmap, int K, int identity_k,
                                                  long long R, const long long *__restrict__ r_dev,
                                                  const float *__restrict__ T, long long t_rows, long long row_pitch,
                                                  long long k_pitch, const float *__restrict__ bias, int C,
                                                  H *__restrict__ Y) {
    __shared__ long long sSrc[128];
    const long long r = blockIdx.x;
    long long Rv = r_dev ? *r_dev : R;
    Rv = Rv < R ? Rv : R;
    if (threadIdx.x < K) {
        const int k = threadIdx.x;
        long long s = -1;
        if (r < Rv) s = (!table || k == identity_k) ? r : (long long)table[(long long)kmap.v[k] * R + r];
        sSrc[k] = (s >= 0 && s < t_rows) ? s : -1;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float v = bias ? bias[c] : 0.f;
        for (int k = 0; k < K; ++k) {
            const long long s = sSrc[k];
            if (s >= 0) v += T[s * row_pitch + k * k_pitch + c];
        }
        wfs_st(Y + r * C + c, v);
    }
}

inline int pad8(int c) { return (c + 7) & ~7; }

template <typename H>
int launch_gemm(const GemmArgs &g, int a_km, int b_km, hipStream_t stream) {
    const long long nblk = (long long)wfs_cdiv(g.M, GT) * wfs_cdiv(g.N, GT) * g.nz;
    if (nblk == 0) return WFS_OK;
    WFS_REQUIRE(nblk < (1ll << 30), WFS_EINVAL, "product of %d x %d x %d tiles is too large", g.M, g.N, g.nz);
    const dim3 grid((unsigned)(wfs_cdiv(nblk, 8) * 8)), block(GTHREADS);
#define WFS_GEMM(AK, BK)                                                                                          \
    do {                                                                                                          \
        static bool attr_set = false;                                                                             \
        if (!attr_set) {                                                                                          \
            WFS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm16<H, AK, BK>),                \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES));          \
            attr_set = true;                                                                                      \
        }                                                                                                         \
        k_gemm16<H, AK, BK><<<grid, block, G_LDS_BYTES, stream>>>(g);                                             \
    } while (0)
    if (!a_km && !b_km) WFS_GEMM(false, false);
    else if (!a_km && b_km) WFS_GEMM(false, true);
    else if (a_km && b_km) WFS_GEMM(true, true);
    else WFS_GEMM(true, false);
#undef WFS_GEMM
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

int gemm(const GemmArgs &g, int a_km, int b_km, int dtype, hipStream_t stream) {
    return dtype == WFS_F16 ? launch_gemm<wfs_f16>(g, a_km, b_km, stream) : launch_gemm<wfs_bf16>(g, a_km, b_km, stream);
}

// carve 256-byte aligned pieces out of the caller's workspace
struct Carver {
    unsigned char *p;
    size_t left;
    void *take(size_t bytes) {
        bytes = wfs_align_up(bytes, 256);
        if (bytes > left) return nullptr;
        void *r = p;
        p += bytes;
        left -= bytes;
        return r;
    }
};

constexpr long long WIDE_MAX_WORKSPACE = 3ll << 30;
bool g_wide_on = true;

// workspace of one wide product: padded filters + (gathered rows | padded rows + per-offset products) + split partials
struct WidePlan {
    bool dense_first;       // source side shorter: dense product over the source rows, then the ordered sum
    int nz, nseg;           // gather-first: the K offsets split over nz batches of nseg segments
    size_t w_bytes, rows_bytes, t_bytes;
};

WidePlan conv_plan(int K, long long R, long long X_rows, int Cx, int Cy, int Cw_in, int Cw_out, bool has_table) {
    WidePlan p;
    p.dense_first = has_table && X_rows < R;
    p.w_bytes = wfs_align_up((size_t)K * Cw_in * pad8(Cw_out) * 2, 256);
    if (p.dense_first) {
        p.nz = K;
        p.nseg = 1;
        p.rows_bytes = wfs_align_up((size_t)X_rows * pad8(Cx) * 2, 256);
        p.t_bytes = wfs_align_up((size_t)X_rows * K * pad8(Cy) * 4, 256);
    } else {
        // enough blocks for the chip: split the offsets over batches while the tiles alone leave CUs idle
        const long long tiles = wfs_cdiv(R, GT) * wfs_cdiv(Cy, GT);
        int nz = 1;
        while (nz < K && tiles * nz < 384) ++nz;
        p.nseg = (int)wfs_cdiv(K, nz);
        p.nz = (int)wfs_cdiv(K, p.nseg);
        p.rows_bytes = wfs_align_up((size_t)R * K * pad8(Cx) * 2, 256);
        p.t_bytes = p.nz > 1 ? wfs_align_up((size_t)p.nz * R * pad8(Cy) * 4, 256) : 0;
    }
    return p;
}

}  // namespace

// which layers take this path: 16-bit rows, one side of the filter at least 256 channels wide (measured against the
// 32 x 32-tile kernels of gather_conv.hip: profiles/r03_microbench_wide.txt), workspace within bounds
extern "C" size_t wfs_wide_conv_workspace_bytes(int32_t K, int64_t R, int64_t X_rows, int32_t Cx, int32_t Cy,
                                                int32_t has_table);

extern "C" int wfs_wide_enable(int32_t on) {
    const int was = g_wide_on ? 1 : 0;
    g_wide_on = on != 0;
    return was;
}

extern "C" int wfs_wide_conv_ok(int32_t K, int64_t R, int64_t X_rows, int32_t Cx, int32_t Cy, int32_t dtype) {
    if (!g_wide_on || (dtype != WFS_BF16 && dtype != WFS_F16)) return 0;
    if (K < 1 || K > 128 || R < 1 || X_rows < 1 || Cx < 8 || Cy < 8) return 0;
    if ((Cx > Cy ? Cx : Cy) < 256) return 0;
    return (long long)wfs_wide_conv_workspace_bytes(K, R, X_rows, Cx, Cy, 1) <= WIDE_MAX_WORKSPACE;
}

extern "C" size_t wfs_wide_conv_workspace_bytes(int32_t K, int64_t R, int64_t X_rows, int32_t Cx, int32_t Cy,
                                                int32_t has_table) {
    // the filter is [Cx][Cy] for the forward product and [Cy][Cx] for dX: padded differently, take the larger
    const WidePlan p = conv_plan(K, R, X_rows, Cx, Cy, Cx, Cy, has_table != 0);
    const WidePlan q = conv_plan(K, R, X_rows, Cx, Cy, Cy, Cx, has_table != 0);
    return (p.w_bytes > q.w_bytes ? p.w_bytes : q.w_bytes) + p.rows_bytes + p.t_bytes + 1024;
}

extern "C" int wfs_wide_gather_conv(const int32_t *table, const int32_t *kmap_host, int32_t K, int32_t identity_k,
                                    int64_t R, const void *X, int64_t X_rows, int32_t Cx, const float *W, int32_t Cw_in,
                                    int32_t Cw_out, int32_t transpose_w, const float *bias, void *Y, int32_t dtype,
                                    const int64_t *r_dev_, void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const long long *r_dev = (const long long *)r_dev_;
    WFS_REQUIRE(dtype == WFS_BF16 || dtype == WFS_F16, WFS_EINVAL, "the wide path takes 16-bit rows (dtype %d)", dtype);
    WFS_REQUIRE(K >= 1 && K <= 128, WFS_EINVAL, "kernel volume %d not in [1,128]", K);
    const int Cy = transpose_w ? Cw_in : Cw_out;
    WFS_REQUIRE(Cx == (transpose_w ? Cw_out : Cw_in), WFS_EINVAL, "channel mismatch: X has %d, filter wants %d", Cx,
                transpose_w ? Cw_out : Cw_in);
    if (R == 0) return WFS_OK;
    WFS_REQUIRE((table || (K == 1 && identity_k == 0)) && X && W && Y && workspace, WFS_EINVAL, "NULL device pointer");
    WFS_REQUIRE(X_rows >= 1, WFS_EINVAL, "no source rows");
    WFS_REQUIRE(table || X_rows == R, WFS_EINVAL, "a product without a table maps row r to row r (%lld vs %lld rows)",
                (long long)X_rows, (long long)R);
    WFS_REQUIRE(R < (1ll << 31) && X_rows < (1ll << 31), WFS_EINVAL, "row count beyond 2^31");
    KMapW km;
    for (int k = 0; k < K; ++k) {
        km.v[k] = kmap_host ? kmap_host[k] : k;
        WFS_REQUIRE(km.v[k] >= 0 && km.v[k] < K, WFS_EINVAL, "kmap[%d] out of range", k);
    }
    const WidePlan p = conv_plan(K, R, X_rows, Cx, Cy, Cw_in, Cw_out, table != nullptr);
    WFS_REQUIRE(workspace_bytes >= p.w_bytes + p.rows_bytes + p.t_bytes, WFS_EWORKSPACE, "workspace %zu < %zu",
                workspace_bytes, p.w_bytes + p.rows_bytes + p.t_bytes);
    WFS_REQUIRE(((uintptr_t)workspace & 15) == 0, WFS_EINVAL, "workspace must be 16-byte aligned");
    WfsTimerScope timer(WFS_TIMER_GATHER_CONV, stream);
    Carver cv{(unsigned char *)workspace, workspace_bytes};
    unsigned short *Wh = (unsigned short *)cv.take(p.w_bytes);
    unsigned short *rows = (unsigned short *)cv.take(p.rows_bytes);
    float *T = p.t_bytes ? (float *)cv.take(p.t_bytes) : nullptr;
    const int Cxp = pad8(Cx), Cyp = pad8(Cy), Cwp = pad8(Cw_out);
    const unsigned short *Xs = (const unsigned short *)X;
    {   // filters -> 16 bit, [K][Cw_in][Cwp]
        const long long wrows = (long long)K * Cw_in, n = wrows * (Cwp >> 3);
        if (dtype == WFS_F16)
            k_pad_f32<wfs_f16><<<dim3((unsigned)wfs_cdiv(n, 256)), 256, 0, stream>>>(W, wrows, Cw_out, Cwp, (wfs_f16 *)Wh);
        else
            k_pad_f32<wfs_bf16><<<dim3((unsigned)wfs_cdiv(n, 256)), 256, 0, stream>>>(W, wrows, Cw_out, Cwp, Wh);
        WFS_LAUNCH_CHECK();
    }
    GemmArgs g{};
    g.B = Wh;
    g.ldb = Cwp;
    g.sB = (long long)Cw_in * Cwp;
    g.Ks = Cx;
    g.nseg_total = K;
    // filter operand: forward W[k] is [contraction Cx][Cy] -> contraction-major; dX W[k] is [Cy][contraction Cx]
    const int b_km = transpose_w ? 0 : 1;
    if (p.dense_first) {
        // rows -> aligned, padded (no gather: identity)
        const long long n = X_rows * (Cxp >> 3);
        k_pad_rows<<<dim3((unsigned)wfs_cdiv(n, 256)), 256, 0, stream>>>(nullptr, km, 1, 0, X_rows, nullptr, Xs, X_rows, Cx,
                                                                       Cxp, rows);
        WFS_LAUNCH_CHECK();
        g.A = rows;
        g.lda = Cxp;
        g.sA = 0;
        g.C = T;
        g.ldc = (long long)K * Cyp;
        g.zC = Cyp;
        g.M = (int)X_rows;
        g.N = Cy;
        g.nseg = 1;
        g.nz = K;
        int rc = gemm(g, 0, b_km, dtype, stream);
        if (rc != WFS_OK) return rc;
        if (dtype == WFS_F16)
            k_sum_rows<wfs_f16><<<dim3((unsigned)R), 256, 0, stream>>>(table, km, K, identity_k, R, r_dev, T, X_rows,
                                                                      (long long)K * Cyp, Cyp, bias, Cy, (wfs_f16 *)Y);
        else
            k_sum_rows<wfs_bf16><<<dim3((unsigned)R), 256, 0, stream>>>(table, km, K, identity_k, R, r_dev, T, X_rows,
                                                                       (long long)K * Cyp, Cyp, bias, Cy, (wfs_bf16 *)Y);
        WFS_LAUNCH_CHECK();
        return WFS_OK;
    }
    {   // gathered rows G[r, (k, c)]
        const long long n = R * K * (Cxp >> 3);
        k_pad_rows<<<dim3((unsigned)wfs_cdiv(n, 256)), 256, 0, stream>>>(table, km, K, identity_k, R, r_dev, Xs, X_rows, Cx,
                                                                       Cxp, rows);
        WFS_LAUNCH_CHECK();
    }
    g.A = rows;
    g.lda = (long long)K * Cxp;
    g.sA = Cxp;
    g.M = (int)R;
    g.N = Cy;
    g.nseg = p.nseg;
    g.nz = p.nz;
    if (p.nz == 1) {
        g.C = Y;
        g.ldc = Cy;
        g.zC = 0;
        g.out_h = 1;
        g.bias = bias;
        return gemm(g, 0, b_km, dtype, stream);
    }
    g.C = T;
    g.ldc = Cyp;
    g.zC = R * (long long)Cyp;
    int rc = gemm(g, 0, b_km, dtype, stream);
    if (rc != WFS_OK) return rc;
    if (dtype == WFS_F16)
        k_sum_rows<wfs_f16><<<dim3((unsigned)R), 256, 0, stream>>>(nullptr, km, p.nz, -1, R, r_dev, T, R, Cyp,
                                                                  R * (long long)Cyp, bias, Cy, (wfs_f16 *)Y);
    else
        k_sum_rows<wfs_bf16><<<dim3((unsigned)R), 256, 0, stream>>>(nullptr, km, p.nz, -1, R, r_dev, T, R, Cyp,
                                                                   R * (long long)Cyp, bias, Cy, (wfs_bf16 *)Y);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

// dW of a wide layer (called from wfs_gather_dw): dW[k][a][b] (swap: dW[k][b][a]) = sum_r S[r][a] G[table[k][r]][b]
size_t wfs_wide_dw_workspace(int K, long long R, int Cs, int Cg) {
    return wfs_align_up((size_t)R * pad8(Cs) * 2, 256) + wfs_align_up((size_t)R * K * pad8(Cg) * 2, 256) + 1024;
}

bool wfs_wide_dw_ok(int K, long long R, int Cs, int Cg, int dtype) {
    if (!g_wide_on || (dtype != WFS_BF16 && dtype != WFS_F16)) return false;
    if (K < 1 || K > 128 || R < 1 || R >= (1ll << 31) || Cs < 8 || Cg < 8 || (Cs > Cg ? Cs : Cg) < 256) return false;
    return (long long)wfs_wide_dw_workspace(K, R, Cs, Cg) <= WIDE_MAX_WORKSPACE;
}

int wfs_launch_wide_dw(const int *table, const int *kmap_host, int K, int identity_k, long long R, const long long *r_dev,
                       const void *S, int Cs, const void *G, long long G_rows, int Cg, int swap, float *dW, int dtype,
                       void *workspace, size_t workspace_bytes, hipStream_t stream) {
    WFS_REQUIRE(workspace_bytes >= wfs_wide_dw_workspace(K, R, Cs, Cg), WFS_EWORKSPACE, "workspace %zu < %zu",
                workspace_bytes, wfs_wide_dw_workspace(K, R, Cs, Cg));
    WFS_REQUIRE(((uintptr_t)workspace & 15) == 0, WFS_EINVAL, "workspace must be 16-byte aligned");
    KMapW km;
    for (int k = 0; k < K; ++k) km.v[k] = kmap_host ? kmap_host[k] : k;
    Carver cv{(unsigned char *)workspace, workspace_bytes};
    const int Csp = pad8(Cs), Cgp = pad8(Cg);
    unsigned short *Sp = (unsigned short *)cv.take((size_t)R * Csp * 2);
    unsigned short *Gg = (unsigned short *)cv.take((size_t)R * K * Cgp * 2);
    WFS_REQUIRE(Sp && Gg, WFS_EWORKSPACE, "workspace too small");
    long long n = R * (Csp >> 3);
    k_pad_rows<<<dim3((unsigned)wfs_cdiv(n, 256)), 256, 0, stream>>>(nullptr, km, 1, 0, R, r_dev, (const unsigned short *)S,
                                                                   R, Cs, Csp, Sp);
    WFS_LAUNCH_CHECK();
    n = R * K * (Cgp >> 3);
    k_pad_rows<<<dim3((unsigned)wfs_cdiv(n, 256)), 256, 0, stream>>>(table, km, K, identity_k, R, r_dev,
                                                                   (const unsigned short *)G, G_rows, Cg, Cgp, Gg);
    WFS_LAUNCH_CHECK();
    GemmArgs g{};
    g.Ks = (int)R;
    g.k_dev = r_dev;
    g.nseg_total = K;
    g.nseg = 1;
    g.nz = K;
    g.C = dW;
    g.zC = (long long)Cs * Cg;
    if (!swap) {
        g.A = Sp, g.lda = Csp, g.sA = 0, g.M = Cs;
        g.B = Gg, g.ldb = (long long)K * Cgp, g.sB = Cgp, g.N = Cg;
        g.ldc = Cg;
    } else {
        g.A = Gg, g.lda = (long long)K * Cgp, g.sA = Cgp, g.M = Cg;
        g.B = Sp, g.ldb = Csp, g.sB = 0, g.N = Cs;
        g.ldc = Cs;
    }
    return gemm(g, 1, 1, dtype, stream);
}
