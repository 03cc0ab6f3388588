// wide.hip -- sparse convolutions with hundreds to thousands of channels as dense matrix-core products.
//
// The reference's hybrid net (BASELINE configs[4]: TemporalConvNet front end -> SparseConv2d 2048 -> 1697 (1 x 1)
// -> 1021 -> 345 (3 x 3), src/models/SPConvBlocks.py:450-516 through spconv's indice_conv) spends its time in
// products of tens of GFLOP per layer.  A 32-row x 32-column block (gather_conv.hip) re-reads 62 MB of filters per
// row tile there; this file carries those layers instead:
//
//   k_gemm16      C[z] = A[z] . B[z]^T on v_mfma_f32_32x32x16_{bf16,f16} (16-bit rows) or v_mfma_f32_32x32x2_f32 (fp32
//                 rows: bitwise an fp32 fma chain, the 1e-5 path): 128 x 128 output tile, 128-byte-deep steps (64 / 32
//                 elements), 4 waves (64 x 64 each), double-buffered LDS filled from two register stages (the global
//                 loads of tile t + 2 in flight under the MFMAs of tiles t and t + 1), fp32 accumulate.  Either operand may be stored contraction-contiguous
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
constexpr int GKB = 128;          // contraction depth of one step in BYTES: 64 16-bit or 32 fp32 elements
constexpr int GTHREADS = 256;     // 4 waves: 2 (rows) x 2 (columns) of 64 x 64
// Timing knock-outs (tools/exp/knock_gemm.sh): -DWFS_GEMM_KNOCK=bits builds a library whose k_gemm16 skips a phase --
// 1 the global loads after the first two tiles, 2 the MFMAs, 4 the LDS stores after the first two tiles.  Results are
// wrong by construction; 0 = nothing.
#ifndef WFS_GEMM_KNOCK
#define WFS_GEMM_KNOCK 0
#endif

struct GemmArgs {
    const void *A, *B;            // elements of the row type (2 or 4 bytes)
    void *C;
    const float *bias;            // [N] or NULL (16-bit output only)
    long long lda, ldb, ldc;      // row pitches in elements (lda, ldb: multiples of 16 bytes)
    long long sA, sB;             // element offset of segment s: A + s * sA, B + s * sB
    long long zC, pC;             // element offsets in C of segment batch zs and of contraction part `part`
    int M, N, Ks;                 // Ks: contraction length of ONE segment
    int nseg_total, nseg;         // segments in all / per batch: batch zs takes segments zs * nseg ...
    int nz;                       // segment batches
    int ksplit, kchunk;           // every segment's contraction is cut into ksplit parts of kchunk (multiple of a step):
                                  // block z = zs * ksplit + part writes its own slab of C (summed by k_sum_rows)
    const long long *k_dev;       // optional device-side contraction length (<= Ks): rows of a dW product
    int out_h;                    // 1: C holds elements of the row type (bias added), 0: fp32 partial products
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

// One operand tile (BO rows of the output dimension x 128 bytes of the contraction) from global memory into BO / 32 x 16
// bytes per thread; ES = element size.  KM == false: stored [row][k]: a thread takes piece (t & 7) of rows (t >> 3) +
// 32 i -- 8 lanes read one full 128-B line.  KM == true: stored [k][row]: 16-byte chunk t % CH of contraction rows
// t / CH + RP i.  Rows / columns beyond the matrix edge are read from a clamped address (their products are never
// stored); contraction indices beyond the segment are read from a clamped address too and zeroed when the tile is
// written to LDS (s_store) -- a select right behind the load would make the wave wait for its prefetch before the MFMAs
// it is meant to run under (measured: s_waitcnt vmcnt(0) ahead of the MFMA block, 41 % of the wave cycles waiting).
template <bool KM, int BO, int ES>
__device__ __forceinline__ void g_load(uint4 (&r)[BO / 32], const char *__restrict__ base, long long ld, int o0, int lim,
                                       int kk0, int Ks, int t) {
    constexpr int PE = 16 / ES;                      // elements per 16-byte piece
    if constexpr (!KM) {
        const int p = t & 7, rr = t >> 3;
        const bool kok = kk0 + p * PE < Ks;
        const int kk = kok ? kk0 + p * PE : 0;
#pragma unroll
        for (int i = 0; i < BO / 32; ++i) {
            int row = o0 + rr + 32 * i;
            row = row < lim ? row : lim - 1;
            r[i] = *reinterpret_cast<const uint4 *>(base + ((long long)row * ld + kk) * ES);
        }
    } else {
        constexpr int CH = BO / PE, RP = GTHREADS / CH;      // chunks per contraction row, rows per pass
        const int ch = t % CH, rr = t / CH;
        int col = o0 + ch * PE;
        col = col < lim ? col : o0;                  // whole chunks beyond the edge: any valid address
#pragma unroll
        for (int i = 0; i < BO / 32; ++i) {
            const int kk = kk0 + rr + RP * i;
            const bool ok = kk < Ks;
            r[i] = *reinterpret_cast<const uint4 *>(base + ((long long)(ok ? kk : 0) * ld + col) * ES);
        }
    }
}

// LDS images (byte offsets inside one operand tile of BO x 128 bytes), every read conflict-free (SQ_LDS_BANK_CONFLICT 0):
//   16-bit [row][k]: 128-B rows, 16-B piece p of row r at r * 128 + ((p ^ ((r >> 1) & 7)) << 4): the 16-lane groups of
//                    ds_read_b128 ({0-3,12-15,20-27}, ...; MI355X_MICROARCH.md "LDS") land on 16 distinct bank groups
//   16-bit [k][row]: 2 BO-byte rows, chunk c of contraction row k at k * 2 BO + ((c ^ ((k & 3) << 2)) << 4): the 4 rows
//                    x 32 columns one half-wave takes with ds_read_b64_tr_b16 cover all 64 banks once
//   fp32 [row][k]:   128-B rows read one dword per lane (row = lane & 31): piece p at p ^ (r & 7), and the four words of
//                    a piece rotated by (r >> 3) & 3 -- 32 rows, 32 banks
//   fp32 [k][row]:   4 BO-byte rows, plain: a half-wave reads 32 consecutive dwords
__device__ __forceinline__ uint4 rot_words(uint4 v, int rot) {       // word j of the result = word (j - rot) & 3 of v
    const unsigned a = v.x, b = v.y, c = v.z, d = v.w;
    uint4 o;
    o.x = rot == 0 ? a : rot == 1 ? d : rot == 2 ? c : b;
    o.y = rot == 0 ? b : rot == 1 ? a : rot == 2 ? d : c;
    o.z = rot == 0 ? c : rot == 1 ? b : rot == 2 ? a : d;
    o.w = rot == 0 ? d : rot == 1 ? c : rot == 2 ? b : a;
    return o;
}

template <bool KM, int BO, int ES>
__device__ __forceinline__ void s_store(unsigned char *tile, const uint4 (&r)[BO / 32], int kk0, int Ks, int t) {
    constexpr int PE = 16 / ES;
    if constexpr (!KM) {
        const int p = t & 7, rr = t >> 3;
        const bool kok = kk0 + p * PE < Ks;
        if constexpr (ES == 2) {
            unsigned char *d = tile + rr * 128 + ((p ^ ((rr >> 1) & 7)) << 4);
#pragma unroll
            for (int i = 0; i < BO / 32; ++i) *reinterpret_cast<uint4 *>(d + i * 32 * 128) = keep_if(r[i], kok);
        } else {
            unsigned char *d = tile + rr * 128 + ((p ^ (rr & 7)) << 4);
            const int rot = (rr >> 3) & 3;           // rows rr + 32 i share it
#pragma unroll
            for (int i = 0; i < BO / 32; ++i)
                *reinterpret_cast<uint4 *>(d + i * 32 * 128) = rot_words(keep_if(r[i], kok), rot);
        }
    } else {
        constexpr int CH = BO / PE, RP = GTHREADS / CH, PITCH = BO * ES;
        static_assert(RP % 4 == 0, "the row swizzle must not change from pass to pass");
        const int ch = t % CH, rr = t / CH;
        unsigned char *d = tile + rr * PITCH + ((ES == 2 ? (ch ^ ((rr & 3) << 2)) : ch) << 4);
#pragma unroll
        for (int i = 0; i < BO / 32; ++i)
            *reinterpret_cast<uint4 *>(d + i * RP * PITCH) = keep_if(r[i], kk0 + rr + RP * i < Ks);
    }
}

// 16-bit MFMA operand (32 rows of the output dimension starting at o, contraction sub-step s of 16) of lane (i = lane &
// 31, h = lane >> 5): elements k = 16 s + 8 h .. + 7 of row o + i.
template <bool KM, int BO>
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
        const unsigned char *a = tile + kk * (2 * BO) + (((col >> 3) ^ ((kk & 3) << 2)) << 4) + 8 * ((col >> 2) & 1);
        typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + 4 * (2 * BO)));
        return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
}

// fp32 MFMA operand (v_mfma_f32_32x32x2_f32): element k = 2 s + h of row o + i
template <bool KM, int BO>
__device__ __forceinline__ float frag32(const unsigned char *tile, int o, int s, int lane) {
    const int row = o + (lane & 31), kk = 2 * s + (lane >> 5);
    if constexpr (!KM)
        return *reinterpret_cast<const float *>(tile + row * 128 + ((((kk >> 2) ^ (row & 7))) << 4) +
                                                (((kk & 3) + (row >> 3)) & 3) * 4);
    else
        return *reinterpret_cast<const float *>(tile + kk * (4 * BO) + row * 4);
}

// BM x BN output tile, 4 waves as 2 (rows) x 2 (columns), each BM / 2 x BN / 2.  Two LDS buffers; DEPTH register
// stages: with DEPTH == 2 the global loads of tile t + 2 are issued before the MFMAs of tile t and written to LDS
// after those of tile t + 1 -- two compute phases to arrive in.
template <typename H, bool A_KM, bool B_KM, int BM, int BN, int DEPTH>
__global__ void __launch_bounds__(GTHREADS) k_gemm16(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int ES = sizeof(H), GK = GKB / ES;
    constexpr int TM = BM / 64, TN = BN / 64;                   // 32 x 32 MFMA tiles of a wave
    constexpr int A_BYTES = BM * GKB, STAGE = (BM + BN) * GKB;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w & 1, wn = w >> 1;
    // consecutive block ids go round the 8 XCDs: give every XCD one contiguous range of tiles, rows fastest, so that
    // the blocks sharing a B panel (the large operand: filters / gathered rows) sit behind one L2
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const long long nblk = (long long)tiles_m * tiles_n * g.nz * g.ksplit;
    const long long per = (nblk + 7) >> 3;
    const long long id = (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (id >= nblk) return;
    const int tm = (int)(id % tiles_m);
    const int tn = (int)((id / tiles_m) % tiles_n);
    const int z = (int)(id / ((long long)tiles_m * tiles_n));
    const int zs = z / g.ksplit, part = z % g.ksplit;
    const int m0 = tm * BM, n0 = tn * BN;
    int Ks = g.Ks;
    if (g.k_dev) {
        const long long kd = *g.k_dev;
        Ks = kd < Ks ? (int)(kd < 0 ? 0 : kd) : Ks;
    }
    const int k_lo = part * g.kchunk;
    const int k_hi = k_lo + g.kchunk < Ks ? k_lo + g.kchunk : Ks;
    const int seg0 = zs * g.nseg;
    const int nseg = g.nseg_total - seg0 < g.nseg ? g.nseg_total - seg0 : g.nseg;
    const int per_seg = k_hi > k_lo ? (k_hi - k_lo + GK - 1) / GK : 0;
    const int nsteps = nseg * per_seg;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    uint4 ra[DEPTH][BM / 32], rb[DEPTH][BN / 32];
    const char *Ab = reinterpret_cast<const char *>(g.A), *Bb = reinterpret_cast<const char *>(g.B);
    auto fetch = [&](int st, uint4 (&xa)[BM / 32], uint4 (&xb)[BN / 32]) {
        if ((WFS_GEMM_KNOCK & 1) && st > 1) return;
        const int seg = seg0 + st / per_seg, kk0 = k_lo + (st % per_seg) * GK;
        g_load<A_KM, BM, ES>(xa, Ab + seg * g.sA * ES, g.lda, m0, g.M, kk0, k_hi, t);
        g_load<B_KM, BN, ES>(xb, Bb + seg * g.sB * ES, g.ldb, n0, g.N, kk0, k_hi, t);
    };
    auto park = [&](int st, const uint4 (&xa)[BM / 32], const uint4 (&xb)[BN / 32]) {      // tile st -> buffer st & 1
        if ((WFS_GEMM_KNOCK & 4) && st > 1) return;
        const int kk0 = k_lo + (st % per_seg) * GK;
        s_store<A_KM, BM, ES>(lds + (st & 1) * STAGE, xa, kk0, k_hi, t);
        s_store<B_KM, BN, ES>(lds + (st & 1) * STAGE + A_BYTES, xb, kk0, k_hi, t);
    };
    auto compute = [&](int buf) {
        const unsigned char *ta = lds + buf * STAGE, *tb = ta + A_BYTES;
        if constexpr (ES == 2) {
#pragma unroll
            for (int s = 0; s < GK / 16; ++s) {
                s16x8 af[TM], bf[TN];
#pragma unroll
                for (int a = 0; a < TM; ++a) af[a] = frag<A_KM, BM>(ta, wm * (BM / 2) + a * 32, s, lane);
#pragma unroll
                for (int b = 0; b < TN; ++b) bf[b] = frag<B_KM, BN>(tb, wn * (BN / 2) + b * 32, s, lane);
                if (WFS_GEMM_KNOCK & 2) {          // no MFMA: keep the fragment reads alive
#pragma unroll
                    for (int a = 0; a < TM; ++a) asm volatile("" ::"v"(af[a]));
#pragma unroll
                    for (int b = 0; b < TN; ++b) asm volatile("" ::"v"(bf[b]));
                    continue;
                }
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) acc[a][b] = mfma16<H>(af[a], bf[b], acc[a][b]);
            }
        } else {
#pragma unroll
            for (int s = 0; s < GK / 2; ++s) {
                float af[TM], bf[TN];
#pragma unroll
                for (int a = 0; a < TM; ++a) af[a] = frag32<A_KM, BM>(ta, wm * (BM / 2) + a * 32, s, lane);
#pragma unroll
                for (int b = 0; b < TN; ++b) bf[b] = frag32<B_KM, BN>(tb, wn * (BN / 2) + b * 32, s, lane);
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
        }
    };
    if constexpr (DEPTH == 1) {
        if (nsteps > 0) {
            fetch(0, ra[0], rb[0]);
            park(0, ra[0], rb[0]);
        }
        __syncthreads();
        for (int st = 0; st < nsteps; ++st) {
            const bool more = st + 1 < nsteps;
            if (more) fetch(st + 1, ra[0], rb[0]);        // in flight under the MFMAs below
            compute(st & 1);
            if (more) park(st + 1, ra[0], rb[0]);         // the other buffer: its readers passed the last barrier
            __syncthreads();
        }
    } else {
        // invariant at the top of step st: tile st is in LDS buffer st & 1, tile st + 1 is on its way into register
        // stage (st + 1) & 1; stage st & 1 is free (its tile was parked) and takes tile st + 2
        if (nsteps > 0) {
            fetch(0, ra[0], rb[0]);
            park(0, ra[0], rb[0]);
        }
        if (nsteps > 1) fetch(1, ra[1], rb[1]);
        __syncthreads();
        auto step = [&](int st, uint4 (&ia)[BM / 32], uint4 (&ib)[BN / 32], uint4 (&pa)[BM / 32], uint4 (&pb)[BN / 32]) {
            if (st + 2 < nsteps) fetch(st + 2, ia, ib);
            compute(st & 1);
            if (st + 1 < nsteps) park(st + 1, pa, pb);
            __syncthreads();
        };
        int st = 0;
        for (; st + 1 < nsteps; st += 2) {
            step(st, ra[0], rb[0], ra[1], rb[1]);
            step(st + 1, ra[1], rb[1], ra[0], rb[0]);
        }
        if (st < nsteps) step(st, ra[0], rb[0], ra[1], rb[1]);
    }
    // C/D map of the 32 x 32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int h = lane >> 5;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int n = n0 + wn * (BN / 2) + b * 32 + (lane & 31);
            if (n >= g.N) continue;
            const float bv = (g.out_h && g.bias) ? g.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + wm * (BM / 2) + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (m >= g.M) continue;
                const long long e = g.zC * zs + g.pC * part + (long long)m * g.ldc + n;
                if (g.out_h)
                    wfs_st(reinterpret_cast<H *>(g.C) + e, acc[a][b][i] + bv);
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

// fp32 [rows, C] -> 16-bit [rows, Cp], zero padded (the filters: rows = K * Cw_in, C = Cw_out).  One thread = one output
// dword (two neighbouring values): consecutive lanes read consecutive 8 bytes and write consecutive 4.
template <typename H>
__global__ void __launch_bounds__(256) k_pad_f32(const float *__restrict__ src, long long rows, int C, int Cp,
                                                 H *__restrict__ dst) {
    const int half = Cp >> 1;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * half) return;
    const int j = (int)(e % half);
    const long long r = e / half;
    const int c = 2 * j;
    const float *s = src + r * C + c;
    const float lo = c < C ? s[0] : 0.f, hi = c + 1 < C ? s[1] : 0.f;
    reinterpret_cast<unsigned *>(dst)[e] = wfs_pack2<H>(lo, hi);
}

// fp32 rows: dst[r, k * Cp + c] = src[row(k, r), c], Cp = C rounded up to 4 (16-byte aligned rows); row(k, r) as in
// k_pad_rows.  One thread = one 16-byte piece of dst.  With no table and K == 1 this is the padded copy of a matrix
// (filters whose output channel count is not a multiple of 4).
__global__ void __launch_bounds__(256) k_pad_rows32(const int *__restrict__ table, KMapW kmap, int K, int identity_k,
                                                    long long R, const long long *__restrict__ r_dev,
                                                    const float *__restrict__ src, long long src_rows, int C, int Cp,
                                                    float *__restrict__ dst) {
    const int pieces = Cp >> 2;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= R * K * pieces) return;
    const int pc = (int)(e % pieces);
    const int k = (int)((e / pieces) % K);
    const long long r = e / ((long long)pieces * K);
    long long Rv = r_dev ? *r_dev : R;
    Rv = Rv < R ? Rv : R;
    long long s = -1;
    if (r < Rv) s = (!table || k == identity_k) ? r : (long long)table[(long long)kmap.v[k] * R + r];
    float4 out = {0.f, 0.f, 0.f, 0.f};
    if (s >= 0 && s < src_rows) {
        const float *q = src + s * C + pc * 4;
        const int n = C - pc * 4;
        out.x = q[0];
        if (n > 1) out.y = q[1];
        if (n > 2) out.z = q[2];
        if (n > 3) out.w = q[3];
    }
    *reinterpret_cast<float4 *>(dst + (r * K + k) * (long long)Cp + pc * 4) = out;
}

// Y[r, c] = bias[c] + sum_k sum_p T[row(k, r) * row_pitch + k * k_pitch + p * p_pitch + c]  in the fixed order k = 0 ..
// K - 1, p = 0 .. P - 1 (fp32), stored as H.  row(k, r) as in k_pad_rows (a missing neighbour contributes nothing).
// One block per output row.  K <= 256.
template <typename H>
__global__ void __launch_bounds__(256) k_sum_rows(const int *__restrict__ table, KMapW kmap, int K, int identity_k,
                                                  long long R, const long long *__restrict__ r_dev,
                                                  const float *__restrict__ T, long long t_rows, long long row_pitch,
                                                  long long k_pitch, int P, long long p_pitch,
                                                  const float *__restrict__ bias, int C, H *__restrict__ Y) {
    __shared__ long long sSrc[256];
    const long long r = blockIdx.x;
    long long Rv = r_dev ? *r_dev : R;
    Rv = Rv < R ? Rv : R;
    if (threadIdx.x < K) {
        const int k = threadIdx.x;
        long long s = -1;
        if (r < Rv) s = (!table || k == identity_k) ? r : (long long)table[(long long)kmap.v[k < 128 ? k : 0] * R + r];
        sSrc[k] = (s >= 0 && s < t_rows) ? s : -1;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float v = bias ? bias[c] : 0.f;
        for (int k = 0; k < K; ++k) {
            const long long s = sSrc[k];
            if (s < 0) continue;
            const float *q = T + s * row_pitch + k * k_pitch + c;
            if (P == 1) {
                v += q[0];
            } else {
                int pp = 0;
                for (; pp + 4 <= P; pp += 4) {          // four loads in flight, added in order
                    const float a0 = q[pp * p_pitch], a1 = q[(pp + 1) * p_pitch], a2 = q[(pp + 2) * p_pitch],
                                a3 = q[(pp + 3) * p_pitch];
                    v = (((v + a0) + a1) + a2) + a3;
                }
                for (; pp < P; ++pp) v += q[pp * p_pitch];
            }
        }
        wfs_st(Y + r * C + c, v);
    }
}

inline int esize(int dtype) { return dtype == WFS_F32 ? 4 : 2; }
inline int padc(int c, int es) {                   // channel count rounded up to whole 16-byte pieces
    const int q = 16 / es;
    return (c + q - 1) / q * q;
}

template <typename H, int BM, int BN, int DEPTH>
int launch_gemm_tile(const GemmArgs &g, int a_km, int b_km, hipStream_t stream) {
    constexpr int LDS = 2 * (BM + BN) * GKB;
    const long long nblk = (long long)wfs_cdiv(g.M, BM) * wfs_cdiv(g.N, BN) * g.nz * g.ksplit;
    if (nblk == 0) return WFS_OK;
    WFS_REQUIRE(nblk < (1ll << 30), WFS_EINVAL, "product of %d x %d x %d tiles is too large", g.M, g.N, g.nz);
    const dim3 grid((unsigned)(wfs_cdiv(nblk, 8) * 8)), block(GTHREADS);
#define WFS_GEMM(AK, BK)                                                                                          \
    do {                                                                                                          \
        static bool attr_set = false;                                                                             \
        if (!attr_set) {                                                                                          \
            WFS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm16<H, AK, BK, BM, BN, DEPTH>), \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS));                  \
            attr_set = true;                                                                                      \
        }                                                                                                         \
        k_gemm16<H, AK, BK, BM, BN, DEPTH><<<grid, block, LDS, stream>>>(g);                                      \
    } while (0)
    if (!a_km && !b_km) WFS_GEMM(false, false);
    else if (!a_km && b_km) WFS_GEMM(false, true);
    else if (a_km && b_km) WFS_GEMM(true, true);
    else WFS_GEMM(true, false);
#undef WFS_GEMM
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

// The product form: 128 x 128 tiles (two blocks per CU), two register stages.  A 256 x 128 tile (one block per CU) and
// a single register stage were measured and lose: profiles/r03_gemm16_tile_and_pipeline_experiments.txt.
template <typename H>
int launch_gemm(const GemmArgs &g, int a_km, int b_km, hipStream_t stream) {
    constexpr int GK = GKB / (int)sizeof(H);
    WFS_REQUIRE(g.ksplit >= 1 && g.kchunk % GK == 0 && (long long)g.ksplit * g.kchunk >= g.Ks, WFS_EINVAL,
                "bad contraction split %d x %d for %d", g.ksplit, g.kchunk, g.Ks);
    return launch_gemm_tile<H, GT, GT, 2>(g, a_km, b_km, stream);
}

int gemm(const GemmArgs &g, int a_km, int b_km, int dtype, hipStream_t stream) {
    if (dtype == WFS_F32) return launch_gemm<float>(g, a_km, b_km, stream);
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
int g_wide_min_channels = 128;      // a side of the filter at least this wide (wfs_wide_enable)

// A launch wants a block per CU: products with fewer output tiles than half the CUs cut the contraction into `ksplit`
// parts of `kchunk` (each at least 2 steps deep), one fp32 slab per part, summed in part order afterwards (a split that
// does not shorten the launch only adds the pass over the slabs: measured, 1.61 -> 1.80 ms per C5 step).
int plan_ksplit(long long blocks, int Ks, int es, int *kchunk) {
    const int GK = GKB / es;
    int ks = 1;
    // fewer 128 x 128 tiles than half the CUs AND a contraction of at least 16 steps: cut it (a short contraction is a
    // ~10-us launch either way, and the pass over the slabs is another launch: GEP's 763-row dW products lost 7 us each)
    // (fp32 steps hold 64 MFMAs of 64 cycles instead of 16 of 32: there two steps already outweigh the extra launch)
    if (blocks * 2 <= 256 && Ks >= (es == 4 ? 2 : 16) * GK) {
        const long long want = 256 / blocks, deep = wfs_cdiv(Ks, 2 * GK);
        ks = (int)(want < deep ? want : deep);
        if (ks < 1) ks = 1;
    }
    const int chunk = (int)(wfs_cdiv(wfs_cdiv(Ks, ks), GK) * GK);
    *kchunk = chunk > 0 ? chunk : GK;
    return (int)wfs_cdiv(Ks > 0 ? Ks : 1, *kchunk);
}

template <typename H>
int launch_sum_rows(const int *table, const KMapW &km, int K, int identity_k, long long R, const long long *r_dev,
                    const float *T, long long t_rows, long long row_pitch, long long k_pitch, int P, long long p_pitch,
                    const float *bias, int C, H *Y, hipStream_t stream) {
    WFS_REQUIRE(K <= 256, WFS_EINVAL, "%d slabs to sum (at most 256)", K);
    k_sum_rows<H><<<dim3((unsigned)R), 256, 0, stream>>>(table, km, K, identity_k, R, r_dev, T, t_rows, row_pitch, k_pitch,
                                                        P, p_pitch, bias, C, Y);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

int sum_rows(const int *table, const KMapW &km, int K, int identity_k, long long R, const long long *r_dev, const float *T,
             long long t_rows, long long row_pitch, long long k_pitch, int P, long long p_pitch, const float *bias, int C,
             void *Y, int dtype, hipStream_t stream) {
    if (dtype == WFS_F32)
        return launch_sum_rows<float>(table, km, K, identity_k, R, r_dev, T, t_rows, row_pitch, k_pitch, P, p_pitch, bias, C,
                                      (float *)Y, stream);
    if (dtype == WFS_F16)
        return launch_sum_rows<wfs_f16>(table, km, K, identity_k, R, r_dev, T, t_rows, row_pitch, k_pitch, P, p_pitch, bias,
                                        C, (wfs_f16 *)Y, stream);
    return launch_sum_rows<wfs_bf16>(table, km, K, identity_k, R, r_dev, T, t_rows, row_pitch, k_pitch, P, p_pitch, bias, C,
                                     (wfs_bf16 *)Y, stream);
}

// fp32 matrix [rows, C] -> the row type, rows padded to whole 16-byte pieces
int pad_f32(const float *src, long long rows, int C, int Cp, void *dst, int dtype, hipStream_t stream) {
    if (dtype == WFS_F32) {
        const long long n = rows * (Cp >> 2);
        if (n == 0) return WFS_OK;
        KMapW km{};
        k_pad_rows32<<<dim3((unsigned)wfs_cdiv(n, 256)), 256, 0, stream>>>(nullptr, km, 1, 0, rows, nullptr, src, rows, C, Cp,
                                                                         (float *)dst);
        WFS_LAUNCH_CHECK();
        return WFS_OK;
    }
    const long long n = rows * (Cp >> 1);
    if (n == 0) return WFS_OK;
    if (dtype == WFS_F16)
        k_pad_f32<wfs_f16><<<dim3((unsigned)wfs_cdiv(n, 256)), 256, 0, stream>>>(src, rows, C, Cp, (wfs_f16 *)dst);
    else
        k_pad_f32<wfs_bf16><<<dim3((unsigned)wfs_cdiv(n, 256)), 256, 0, stream>>>(src, rows, C, Cp, (wfs_bf16 *)dst);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

// rows of the row type -> 16-byte aligned, zero-padded rows, gathered through the table when given
int pad_rows(const int *table, const KMapW &km, int K, int identity_k, long long R, const long long *r_dev,
             const void *src, long long src_rows, int C, int Cp, void *dst, int dtype, hipStream_t stream) {
    const long long n = R * K * (Cp / (16 / esize(dtype)));
    if (n == 0) return WFS_OK;
    if (dtype == WFS_F32)
        k_pad_rows32<<<dim3((unsigned)wfs_cdiv(n, 256)), 256, 0, stream>>>(table, km, K, identity_k, R, r_dev,
                                                                         (const float *)src, src_rows, C, Cp, (float *)dst);
    else
        k_pad_rows<<<dim3((unsigned)wfs_cdiv(n, 256)), 256, 0, stream>>>(table, km, K, identity_k, R, r_dev,
                                                                       (const unsigned short *)src, src_rows, C, Cp,
                                                                       (unsigned short *)dst);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

// fp32 data whose rows are already whole 16-byte pieces at a 16-byte aligned address need no copy
inline bool f32_in_place(const void *p, int C, int dtype) { return dtype == WFS_F32 && C % 4 == 0 && ((uintptr_t)p & 15) == 0; }

// workspace of one wide product: padded filters + (gathered rows | padded rows + per-offset products) + split partials
struct WidePlan {
    bool dense_first;       // source side shorter: dense product over the source rows, then the ordered sum
    int nz, nseg;           // gather-first: the K offsets split over nz batches of nseg segments
    int ksplit, kchunk;     // contraction split of every segment
    size_t w_bytes, rows_bytes, t_bytes;
};

WidePlan conv_plan(int K, long long R, long long X_rows, int Cx, int Cy, int Cw_in, int Cw_out, bool has_table, int es) {
    WidePlan p;
    p.dense_first = has_table && X_rows < R;
    p.w_bytes = wfs_align_up((size_t)K * Cw_in * padc(Cw_out, es) * es, 256);
    if (p.dense_first) {
        p.nz = K;
        p.nseg = 1;
        p.ksplit = plan_ksplit(wfs_cdiv(X_rows, GT) * wfs_cdiv(Cy, GT) * K, Cx, es, &p.kchunk);
        p.rows_bytes = wfs_align_up((size_t)X_rows * padc(Cx, es) * es, 256);
        p.t_bytes = wfs_align_up((size_t)p.ksplit * X_rows * K * padc(Cy, 4) * 4, 256);
    } else {
        // enough blocks for the chip: split the offsets over batches while the tiles alone leave CUs idle
        const long long tiles = wfs_cdiv(R, GT) * wfs_cdiv(Cy, GT);
        int nz = 1;
        while (nz < K && tiles * nz < 384) ++nz;
        p.nseg = (int)wfs_cdiv(K, nz);
        p.nz = (int)wfs_cdiv(K, p.nseg);
        p.ksplit = plan_ksplit(tiles * p.nz, Cx, es, &p.kchunk);
        p.rows_bytes = wfs_align_up((size_t)R * K * padc(Cx, es) * es, 256);
        p.t_bytes = p.nz * p.ksplit > 1 ? wfs_align_up((size_t)p.nz * p.ksplit * R * padc(Cy, 4) * 4, 256) : 0;
    }
    return p;
}

}  // namespace

extern "C" size_t wfs_wide_conv_workspace_bytes(int32_t K, int64_t R, int64_t X_rows, int32_t Cx, int32_t Cy,
                                                int32_t has_table, int32_t dtype) {
    // the filter is [Cx][Cy] for the forward product and [Cy][Cx] for dX: padded differently, take the larger
    const int es = esize(dtype);
    const WidePlan p = conv_plan(K, R, X_rows, Cx, Cy, Cx, Cy, has_table != 0, es);
    const WidePlan q = conv_plan(K, R, X_rows, Cx, Cy, Cy, Cx, has_table != 0, es);
    return (p.w_bytes > q.w_bytes ? p.w_bytes : q.w_bytes) + p.rows_bytes + p.t_bytes + 1024;
}

extern "C" int wfs_wide_enable(int32_t on) {
    const int was = g_wide_on ? g_wide_min_channels : 0;
    g_wide_on = on != 0;
    if (on >= 8) g_wide_min_channels = on;            // benchmarks: the channel threshold itself
    return was;
}

// which layers take this path: one side of the filter at least 128 channels wide (measured against the 32 x 32-tile
// kernels of gather_conv.hip: profiles/r03_wide_threshold_ab.txt), workspace within bounds
extern "C" int wfs_wide_conv_ok(int32_t K, int64_t R, int64_t X_rows, int32_t Cx, int32_t Cy, int32_t dtype) {
    if (!g_wide_on || !wfs_dtype_ok(dtype)) return 0;
    if (K < 1 || K > 128 || R < 1 || X_rows < 1 || Cx < 8 || Cy < 8) return 0;
    if (R >= (1ll << 31) || X_rows >= (1ll << 31)) return 0;
    if ((Cx > Cy ? Cx : Cy) < g_wide_min_channels) return 0;
    return (long long)wfs_wide_conv_workspace_bytes(K, R, X_rows, Cx, Cy, 1, dtype) <= WIDE_MAX_WORKSPACE;
}

extern "C" size_t wfs_wide_filters_bytes(int32_t K, int32_t Cw_in, int32_t Cw_out, int32_t dtype) {
    const int es = esize(dtype);
    return wfs_align_up((size_t)K * Cw_in * padc(Cw_out, es) * es, 256);
}

extern "C" int wfs_wide_filters(const float *W, int32_t K, int32_t Cw_in, int32_t Cw_out, int32_t dtype, void *Wp,
                                void *stream) {
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    WFS_REQUIRE(K >= 1 && Cw_in >= 1 && Cw_out >= 1 && W && Wp, WFS_EINVAL, "bad filter block");
    WFS_REQUIRE(((uintptr_t)Wp & 15) == 0, WFS_EINVAL, "the filter copy must be 16-byte aligned");
    return pad_f32(W, (long long)K * Cw_in, Cw_out, padc(Cw_out, esize(dtype)), Wp, dtype, (hipStream_t)stream);
}

extern "C" int wfs_wide_gather_conv(const int32_t *table, const int32_t *kmap_host, int32_t K, int32_t identity_k,
                                    int64_t R, const void *X, int64_t X_rows, int32_t Cx, const float *W,
                                    const void *Wp_, int32_t Cw_in, int32_t Cw_out, int32_t transpose_w,
                                    const float *bias, void *Y, int32_t dtype, const int64_t *r_dev_, void *workspace,
                                    size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const long long *r_dev = (const long long *)r_dev_;
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    WFS_REQUIRE(K >= 1 && K <= 128, WFS_EINVAL, "kernel volume %d not in [1,128]", K);
    const int Cy = transpose_w ? Cw_in : Cw_out;
    WFS_REQUIRE(Cx == (transpose_w ? Cw_out : Cw_in), WFS_EINVAL, "channel mismatch: X has %d, filter wants %d", Cx,
                transpose_w ? Cw_out : Cw_in);
    if (R == 0) return WFS_OK;
    WFS_REQUIRE((table || (K == 1 && identity_k == 0)) && X && (W || Wp_) && Y && workspace, WFS_EINVAL,
                "NULL device pointer");
    WFS_REQUIRE(((uintptr_t)Wp_ & 15) == 0, WFS_EINVAL, "the filter copy must be 16-byte aligned");
    WFS_REQUIRE(X_rows >= 1, WFS_EINVAL, "no source rows");
    WFS_REQUIRE(table || X_rows == R, WFS_EINVAL, "a product without a table maps row r to row r (%lld vs %lld rows)",
                (long long)X_rows, (long long)R);
    WFS_REQUIRE(R < (1ll << 31) && X_rows < (1ll << 31), WFS_EINVAL, "row count beyond 2^31");
    KMapW km;
    for (int k = 0; k < K; ++k) {
        km.v[k] = kmap_host ? kmap_host[k] : k;
        WFS_REQUIRE(km.v[k] >= 0 && km.v[k] < K, WFS_EINVAL, "kmap[%d] out of range", k);
    }
    const int es = esize(dtype);
    const WidePlan p = conv_plan(K, R, X_rows, Cx, Cy, Cw_in, Cw_out, table != nullptr, es);
    WFS_REQUIRE(workspace_bytes >= p.w_bytes + p.rows_bytes + p.t_bytes, WFS_EWORKSPACE, "workspace %zu < %zu",
                workspace_bytes, p.w_bytes + p.rows_bytes + p.t_bytes);
    WFS_REQUIRE(((uintptr_t)workspace & 15) == 0, WFS_EINVAL, "workspace must be 16-byte aligned");
    WfsTimerScope timer(WFS_TIMER_GATHER_CONV, stream);
    Carver cv{(unsigned char *)workspace, workspace_bytes};
    void *Wh = cv.take(p.w_bytes);
    void *rows = cv.take(p.rows_bytes);
    float *T = p.t_bytes ? (float *)cv.take(p.t_bytes) : nullptr;
    const int Cxp = padc(Cx, es), Cyp = padc(Cy, 4), Cwp = padc(Cw_out, es);
    int rc = WFS_OK;
    const void *Wp = Wp_;
    if (!Wp) {      // filters in the row type, [K][Cw_in][Cwp]
        if (f32_in_place(W, Cw_out, dtype)) {
            Wp = W;
        } else {
            rc = pad_f32(W, (long long)K * Cw_in, Cw_out, Cwp, Wh, dtype, stream);
            Wp = Wh;
        }
    }
    if (rc != WFS_OK) return rc;
    GemmArgs g{};
    g.B = Wp;
    g.ldb = Cwp;
    g.sB = (long long)Cw_in * Cwp;
    g.Ks = Cx;
    g.nseg_total = K;
    g.ksplit = p.ksplit;
    g.kchunk = p.kchunk;
    // filter operand: forward W[k] is [contraction Cx][Cy] -> contraction-major; dX W[k] is [Cy][contraction Cx]
    const int b_km = transpose_w ? 0 : 1;
    if (p.dense_first) {
        const void *Xa = X;                           // aligned, padded rows
        if (!f32_in_place(X, Cx, dtype)) {
            rc = pad_rows(nullptr, km, 1, 0, X_rows, nullptr, X, X_rows, Cx, Cxp, rows, dtype, stream);
            if (rc != WFS_OK) return rc;
            Xa = rows;
        }
        const long long slab = X_rows * (long long)K * Cyp;
        g.A = Xa, g.lda = Cxp, g.sA = 0;
        g.C = T, g.ldc = (long long)K * Cyp, g.zC = Cyp, g.pC = slab;
        g.M = (int)X_rows, g.N = Cy, g.nseg = 1, g.nz = K;
        rc = gemm(g, 0, b_km, dtype, stream);
        if (rc != WFS_OK) return rc;
        return sum_rows(table, km, K, identity_k, R, r_dev, T, X_rows, (long long)K * Cyp, Cyp, p.ksplit, slab, bias, Cy, Y,
                        dtype, stream);
    }
    if (!table && f32_in_place(X, Cx, dtype) && !r_dev) {
        g.A = X;                                      // a dense fp32 product on rows that need no copy
    } else {
        rc = pad_rows(table, km, K, identity_k, R, r_dev, X, X_rows, Cx, Cxp, rows, dtype, stream);   // G[r, (k, c)]
        if (rc != WFS_OK) return rc;
        g.A = rows;
    }
    g.lda = (long long)K * Cxp, g.sA = Cxp;
    g.M = (int)R, g.N = Cy, g.nseg = p.nseg, g.nz = p.nz;
    if (p.nz * p.ksplit == 1) {
        g.C = Y, g.ldc = Cy, g.out_h = 1, g.bias = bias;
        return gemm(g, 0, b_km, dtype, stream);
    }
    const long long slab = R * (long long)Cyp;
    g.C = T, g.ldc = Cyp, g.zC = p.ksplit * slab, g.pC = slab;
    rc = gemm(g, 0, b_km, dtype, stream);
    if (rc != WFS_OK) return rc;
    return sum_rows(nullptr, km, p.nz * p.ksplit, -1, R, r_dev, T, R, Cyp, slab, 1, 0, bias, Cy, Y, dtype, stream);
}

// dW of a wide layer (called from wfs_gather_dw): dW[k][a][b] (swap: dW[k][b][a]) = sum_r S[r][a] G[table[k][r]][b]
static int dw_split(int K, long long R, int Cs, int Cg, int es, int *kchunk) {
    return plan_ksplit(wfs_cdiv(Cs, GT) * wfs_cdiv(Cg, GT) * K, (int)R, es, kchunk);
}

size_t wfs_wide_dw_workspace(int K, long long R, int Cs, int Cg, int dtype) {
    const int es = esize(dtype);
    int kchunk;
    const int ks = dw_split(K, R, Cs, Cg, es, &kchunk);
    return wfs_align_up((size_t)R * padc(Cs, es) * es, 256) + wfs_align_up((size_t)R * K * padc(Cg, es) * es, 256) +
           (ks > 1 ? wfs_align_up((size_t)ks * K * Cs * Cg * 4, 256) : 0) + 1024;
}

bool wfs_wide_dw_ok(int K, long long R, int Cs, int Cg, int dtype) {
    if (!g_wide_on || !wfs_dtype_ok(dtype)) return false;
    if (K < 1 || K > 128 || R < 1 || R >= (1ll << 31) || Cs < 8 || Cg < 8 || (Cs > Cg ? Cs : Cg) < g_wide_min_channels)
        return false;
    return (long long)wfs_wide_dw_workspace(K, R, Cs, Cg, dtype) <= WIDE_MAX_WORKSPACE;
}

// A^T . B over the rows of two contraction-major operands: C[z][m][n] = sum_r A[r][z sA + m] B[r][z sB + n]
static int rows_product(const void *A, long long lda, long long sA, int M, const void *B, long long ldb, long long sB,
                        int N, long long R, const long long *r_dev, int nz, float *C, float *slabs, int dtype,
                        hipStream_t stream) {
    GemmArgs g{};
    g.A = A, g.lda = lda, g.sA = sA, g.M = M;
    g.B = B, g.ldb = ldb, g.sB = sB, g.N = N;
    g.Ks = (int)R;
    g.k_dev = r_dev;
    g.nseg_total = nz, g.nseg = 1, g.nz = nz;
    g.ksplit = plan_ksplit(wfs_cdiv(M, GT) * wfs_cdiv(N, GT) * nz, (int)R, esize(dtype), &g.kchunk);
    g.ldc = N;
    g.zC = (long long)M * N;
    if (g.ksplit == 1) {
        g.C = C;
        return gemm(g, 1, 1, dtype, stream);
    }
    WFS_REQUIRE(slabs, WFS_EWORKSPACE, "no room for the split partials");
    g.C = slabs;
    g.pC = (long long)nz * M * N;
    int rc = gemm(g, 1, 1, dtype, stream);
    if (rc != WFS_OK) return rc;
    KMapW km{};
    return launch_sum_rows<float>(nullptr, km, 1, -1, (long long)nz * M, nullptr, slabs, (long long)nz * M, N, 0, g.ksplit,
                                  g.pC, nullptr, N, C, stream);
}

int wfs_launch_wide_dw(const int *table, const int *kmap_host, int K, int identity_k, long long R, const long long *r_dev,
                       const void *S, int Cs, const void *G, long long G_rows, int Cg, int swap, float *dW, int dtype,
                       void *workspace, size_t workspace_bytes, hipStream_t stream) {
    WFS_REQUIRE(workspace_bytes >= wfs_wide_dw_workspace(K, R, Cs, Cg, dtype), WFS_EWORKSPACE, "workspace %zu < %zu",
                workspace_bytes, wfs_wide_dw_workspace(K, R, Cs, Cg, dtype));
    WFS_REQUIRE(((uintptr_t)workspace & 15) == 0, WFS_EINVAL, "workspace must be 16-byte aligned");
    KMapW km;
    for (int k = 0; k < K; ++k) km.v[k] = kmap_host ? kmap_host[k] : k;
    Carver cv{(unsigned char *)workspace, workspace_bytes};
    const int es = esize(dtype);
    const int Csp = padc(Cs, es), Cgp = padc(Cg, es);
    int kchunk;
    const int ks = dw_split(K, R, Cs, Cg, es, &kchunk);
    void *Sp = cv.take((size_t)R * Csp * es);
    void *Gg = cv.take((size_t)R * K * Cgp * es);
    float *slabs = ks > 1 ? (float *)cv.take((size_t)ks * K * Cs * Cg * 4) : nullptr;
    WFS_REQUIRE(Sp && Gg && (ks == 1 || slabs), WFS_EWORKSPACE, "workspace too small");
    const void *Sa = S;
    int rc;
    if (!f32_in_place(S, Cs, dtype)) {            // the contraction stops at the valid count: no need to zero rows
        rc = pad_rows(nullptr, km, 1, 0, R, r_dev, S, R, Cs, Csp, Sp, dtype, stream);
        if (rc != WFS_OK) return rc;
        Sa = Sp;
    }
    rc = pad_rows(table, km, K, identity_k, R, r_dev, G, G_rows, Cg, Cgp, Gg, dtype, stream);
    if (rc != WFS_OK) return rc;
    if (!swap)
        return rows_product(Sa, Csp, 0, Cs, Gg, (long long)K * Cgp, Cgp, Cg, R, r_dev, K, dW, slabs, dtype, stream);
    return rows_product(Gg, (long long)K * Cgp, Cgp, Cg, Sa, Csp, 0, Cs, R, r_dev, K, dW, slabs, dtype, stream);
}

// ------------------------------------------------------------------------------------------ dense linear layer
// y = x W^T + b with x [B, I] in the row type, fp32 W [O, I] (torch.nn.Linear's layout), fp32 y: the hybrid net's head
// (Linear(24150, 269), reference src/models/SPConvNet.py:40-52 through LinearBlock) is a 3.3-GFLOP product that streams
// 26 MB of weights: 2 x 3 output tiles, so the contraction is cut into parts (plan_ksplit) and summed in order.
static int lin_split(long long B, int I, int O, int es, int *kchunk) {
    return plan_ksplit(wfs_cdiv(B, GT) * wfs_cdiv(O, GT), I, es, kchunk);
}

extern "C" size_t wfs_wide_linear_workspace_bytes(int64_t B, int32_t I, int32_t O, int32_t dtype) {
    const int es = esize(dtype);
    int kchunk;
    const int ks = lin_split(B, I, O, es, &kchunk);
    const size_t x = wfs_align_up((size_t)B * padc(I, es) * es, 256), w = wfs_align_up((size_t)O * padc(I, es) * es, 256);
    const size_t g = wfs_align_up((size_t)B * padc(O, es) * es, 256);
    const size_t part = wfs_align_up((size_t)ks * B * padc(O, 4) * 4, 256);
    int kc2;
    const int ks2 = plan_ksplit(wfs_cdiv(O, GT) * wfs_cdiv(I, GT), (int)B, es, &kc2);
    const size_t part2 = ks2 > 1 ? wfs_align_up((size_t)ks2 * O * I * 4, 256) : 0;
    return x + w + g + (part > part2 ? part : part2) + 1024;
}

extern "C" int wfs_wide_linear_ok(int64_t B, int32_t I, int32_t O, int32_t dtype) {
    if (!g_wide_on || !wfs_dtype_ok(dtype)) return 0;
    if (B < 1 || B >= (1ll << 31) || I < 256 || O < 9) return 0;
    return (long long)wfs_wide_linear_workspace_bytes(B, I, O, dtype) <= WIDE_MAX_WORKSPACE;
}

extern "C" int wfs_wide_linear_fwd(const void *X, int64_t B, int32_t I, const float *W, const float *bias, int32_t O,
                                   float *Y, int32_t dtype, void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    WFS_REQUIRE(B >= 1 && B < (1ll << 31) && I >= 8 && O >= 1, WFS_EINVAL, "bad shape %lld x %d -> %d", (long long)B, I, O);
    WFS_REQUIRE(X && W && Y && workspace, WFS_EINVAL, "NULL device pointer");
    WFS_REQUIRE(workspace_bytes >= wfs_wide_linear_workspace_bytes(B, I, O, dtype), WFS_EWORKSPACE, "workspace %zu < %zu",
                workspace_bytes, wfs_wide_linear_workspace_bytes(B, I, O, dtype));
    WFS_REQUIRE(((uintptr_t)workspace & 15) == 0, WFS_EINVAL, "workspace must be 16-byte aligned");
    const int es = esize(dtype);
    const int Ip = padc(I, es), Op = padc(O, es), Of = padc(O, 4);
    Carver cv{(unsigned char *)workspace, workspace_bytes};
    void *Xp = cv.take((size_t)B * Ip * es);
    void *Wh = cv.take((size_t)O * Ip * es);
    cv.take((size_t)B * Op * es);
    KMapW km{};
    int rc;
    const void *Xa = X, *Wa = W;
    if (!f32_in_place(X, I, dtype)) {
        rc = pad_rows(nullptr, km, 1, 0, B, nullptr, X, B, I, Ip, Xp, dtype, stream);
        if (rc != WFS_OK) return rc;
        Xa = Xp;
    }
    if (!f32_in_place(W, I, dtype)) {
        rc = pad_f32(W, O, I, Ip, Wh, dtype, stream);
        if (rc != WFS_OK) return rc;
        Wa = Wh;
    }
    GemmArgs g{};
    g.A = Xa, g.lda = Ip, g.M = (int)B;
    g.B = Wa, g.ldb = Ip, g.N = O;
    g.Ks = I;
    g.nseg_total = 1, g.nseg = 1, g.nz = 1;
    g.ksplit = lin_split(B, I, O, es, &g.kchunk);
    float *part = (float *)cv.take((size_t)g.ksplit * B * Of * 4);
    WFS_REQUIRE(part, WFS_EWORKSPACE, "workspace too small");
    g.C = part, g.ldc = Of, g.pC = B * (long long)Of;
    rc = gemm(g, 0, 0, dtype, stream);
    if (rc != WFS_OK) return rc;
    return launch_sum_rows<float>(nullptr, km, 1, -1, B, nullptr, part, B, Of, 0, g.ksplit, g.pC, bias, O, Y, stream);
}

// dX [B, I] (row type, may be NULL), dW [O, I] and db [O] (fp32, may be NULL) from fp32 dY [B, O]
extern "C" int wfs_wide_linear_bwd(const void *X, const float *dY, int64_t B, int32_t I, const float *W, int32_t O,
                                   void *dX, float *dW, float *db, int32_t dtype, void *workspace,
                                   size_t workspace_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(wfs_dtype_ok(dtype), WFS_EINVAL, "bad dtype %d", dtype);
    WFS_REQUIRE(B >= 1 && B < (1ll << 31) && I >= 8 && O >= 1, WFS_EINVAL, "bad shape %lld x %d -> %d", (long long)B, I, O);
    WFS_REQUIRE(X && dY && W && workspace, WFS_EINVAL, "NULL device pointer");
    WFS_REQUIRE(workspace_bytes >= wfs_wide_linear_workspace_bytes(B, I, O, dtype), WFS_EWORKSPACE, "workspace %zu < %zu",
                workspace_bytes, wfs_wide_linear_workspace_bytes(B, I, O, dtype));
    WFS_REQUIRE(((uintptr_t)workspace & 15) == 0, WFS_EINVAL, "workspace must be 16-byte aligned");
    const int es = esize(dtype);
    const int Ip = padc(I, es), Op = padc(O, es);
    Carver cv{(unsigned char *)workspace, workspace_bytes};
    void *Xp = cv.take((size_t)B * Ip * es);
    void *Wh = cv.take((size_t)O * Ip * es);
    void *Gh = cv.take((size_t)B * Op * es);
    KMapW km{};
    int rc;
    const void *Ga = dY;                                               // dY in the row type, [B][Op]
    if (!f32_in_place(dY, O, dtype)) {
        rc = pad_f32(dY, B, O, Op, Gh, dtype, stream);
        if (rc != WFS_OK) return rc;
        Ga = Gh;
    }
    if (dX) {
        const void *Wa = W;
        if (!f32_in_place(W, I, dtype)) {
            rc = pad_f32(W, O, I, Ip, Wh, dtype, stream);
            if (rc != WFS_OK) return rc;
            Wa = Wh;
        }
        GemmArgs g{};                                                  // dX = dY . W: contraction over O
        g.A = Ga, g.lda = Op, g.M = (int)B;
        g.B = Wa, g.ldb = Ip, g.N = I;                                 // W as [k = O][n = I]: contraction-major
        g.Ks = O;
        const int GK = GKB / es;
        g.nseg_total = 1, g.nseg = 1, g.nz = 1, g.ksplit = 1, g.kchunk = (int)(wfs_cdiv(O, GK) * GK);
        g.C = dX, g.ldc = I, g.out_h = 1;
        rc = gemm(g, 0, 1, dtype, stream);
        if (rc != WFS_OK) return rc;
    }
    if (dW) {
        const void *Xa = X;
        if (!f32_in_place(X, I, dtype)) {
            rc = pad_rows(nullptr, km, 1, 0, B, nullptr, X, B, I, Ip, Xp, dtype, stream);
            if (rc != WFS_OK) return rc;
            Xa = Xp;
        }
        int kc;
        const int ks = plan_ksplit(wfs_cdiv(O, GT) * wfs_cdiv(I, GT), (int)B, es, &kc);
        float *slabs = ks > 1 ? (float *)cv.take((size_t)ks * O * I * 4) : nullptr;
        rc = rows_product(Ga, Op, 0, O, Xa, Ip, 0, I, B, nullptr, 1, dW, slabs, dtype, stream);   // dW = dY^T . X
        if (rc != WFS_OK) return rc;
    }
    if (db) {
        // column sums of dY in fp32, fixed order: one block walks the rows
        return launch_sum_rows<float>(nullptr, km, 1, -1, 1, nullptr, dY, 1, 0, 0, (int)B, O, nullptr, O, db, stream);
    }
    return WFS_OK;
}
