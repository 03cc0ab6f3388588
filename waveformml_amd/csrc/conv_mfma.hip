// conv_mfma.hip -- shape-specialised kernels for the PSD net's layers (fp32 storage, exact fp32 MFMA).
//
//   k_gconv32_f32   Cin = Cout = 32 gather conv (forward, and dX with the transposed filter):
//                   one wave = 32 output rows x 32 channels, v_mfma_f32_32x32x2_f32 (exact fp32 fma
//                   chain, MI355X_MICROARCH.md "Matrix cores"), all K filters resident in LDS in
//                   fragment order, per-tile skip of kernel offsets no row of the tile uses.
//   k_gconv_c2c32   Cin = 2 -> Cout = 32 first layer (VALU; 8 lanes per row -> 1 KiB coalesced stores).
//   k_gdw32_f32     dW for 32 x 32 channels: rows are the MFMA K dimension, S and gathered G rows are read
//                   as whole 128-B lines, block-level LDS reduction, deterministic slab reduce.
//   k_gdw_c2c32     dW for the first layer.
//
// Contraction-index trick: the k order inside an MFMA chain is free as long as A and B agree, so lane
// (r, h = lane>>5) feeds channels h*16 .. h*16+15 of its gathered row (one 64-B contiguous read) instead of
// the natural even/odd interleave.
#include <stdlib.h>

#include "wfs_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int F32_GROUP = 4;       // offsets whose row gathers a wave of k_gconv32_f32 has in flight together (64 VGPRs)

struct KMap {
    int v[128];
};

// R = capacity (strides, grid); the number of valid rows comes from device memory when r_dev is given
__device__ __forceinline__ long long valid_rows(long long R, const long long *r_dev) {
    long long v = r_dev ? *r_dev : R;
    return v < R ? v : R;
}

// Packed by-input tables (evconv.hip; include/wfsparse.h "packed tables"): where the kernel is no longer than the
// stride along the last dimension an input row reaches at most ONE output cell per leading offset q, so the table is
// [K / pk, R] and an entry e >= 0 means "row e >> 3, at offset k = q * pk + (e & 7)".  pk == 0: the dense [K, R] form.
__device__ __forceinline__ int packed_entry(int e, int o) { return (e >= 0 && (e & 7) == o) ? (e >> 3) : -1; }
// which table row serves offset k, and which packed offset it must carry (-1: dense table, the entry is the row);
// wave-uniform, computed once per kernel -- the loads themselves stay unconditional
__device__ __forceinline__ void table_row_of(int pk, int k, int *trow, int *osel) {
    const int kq = pk ? k / pk : k;
    *trow = kq;
    *osel = pk ? k - kq * pk : -1;
}
__device__ __forceinline__ int table_value(int e, int osel) { return osel < 0 ? e : packed_entry(e, osel); }

typedef __bf16 bf16x8e __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- fp32 numbers as three bf16 pieces (round 4; the whole story is at k_gdw32_split below)
// x[0..3] -> the three pieces, each as 4 packed bf16 (element i in bits 16 (i & 1) of word i >> 1).  Finite inputs only:
// for +-Inf the first residual is Inf - Inf = NaN (an fp32 product would stay +-Inf); fp32 denormals give denormal pieces
__device__ __forceinline__ void split3_bf16(f32x4 x, bool keep, uint2 &p1, uint2 &p2, uint2 &p3) {
    unsigned u[4], v[4], w[4];
#pragma unroll
    for (int i = 0; i < 4; i += 2) {        // two at a time: the subtractions are v_pk_add_f32
        const float xa = x[i], xb = x[i + 1];      // (a bit_cast straight from a vector element reads element 0 with this compiler)
        u[i] = keep ? __builtin_bit_cast(unsigned, xa) : 0u;
        u[i + 1] = keep ? __builtin_bit_cast(unsigned, xb) : 0u;
        const f32x2 full = {__builtin_bit_cast(float, u[i]), __builtin_bit_cast(float, u[i + 1])};
        const f32x2 top = {__builtin_bit_cast(float, u[i] & 0xFFFF0000u), __builtin_bit_cast(float, u[i + 1] & 0xFFFF0000u)};
        const f32x2 r = full - top;
        const float ra = r[0], rb = r[1];
        v[i] = __builtin_bit_cast(unsigned, ra);
        v[i + 1] = __builtin_bit_cast(unsigned, rb);
        const f32x2 top2 = {__builtin_bit_cast(float, v[i] & 0xFFFF0000u), __builtin_bit_cast(float, v[i + 1] & 0xFFFF0000u)};
        const f32x2 r2 = r - top2;
        const float sa = r2[0], sb = r2[1];
        w[i] = __builtin_bit_cast(unsigned, sa);
        w[i + 1] = __builtin_bit_cast(unsigned, sb);
    }
    // v_perm_b32: bytes 3, 2 of the second operand below bytes 3, 2 of the first
    p1 = uint2{__builtin_amdgcn_perm(u[1], u[0], 0x07060302u), __builtin_amdgcn_perm(u[3], u[2], 0x07060302u)};
    p2 = uint2{__builtin_amdgcn_perm(v[1], v[0], 0x07060302u), __builtin_amdgcn_perm(v[3], v[2], 0x07060302u)};
    p3 = uint2{__builtin_amdgcn_perm(w[1], w[0], 0x07060302u), __builtin_amdgcn_perm(w[3], w[2], 0x07060302u)};
}

// 8 floats (two f32x4: elements 0..3, 4..7) -> the three pieces as MFMA operands of 8 bf16
__device__ __forceinline__ void split3_bf16x8(f32x4 lo, f32x4 hi, uint4 &p1, uint4 &p2, uint4 &p3) {
    uint2 a1, a2, a3, b1, b2, b3;
    split3_bf16(lo, true, a1, a2, a3);
    split3_bf16(hi, true, b1, b2, b3);
    p1 = uint4{a1.x, a1.y, b1.x, b1.y};
    p2 = uint4{a2.x, a2.y, b2.x, b2.y};
    p3 = uint4{a3.x, a3.y, b3.x, b3.y};
}

// ------------------------------------------------------------------------------------------ 32 -> 32, fp32, 16-row tiles
// k_gconv32_f32 gives every wave ONE 32-row tile, and a launch then lasts as long as its heaviest tile: 16 dependent
// v_mfma_f32_32x32x2_f32 (1024 cycles) per active offset, up to K = 27 of them, on a SIMD shared with two other waves --
// the matrix cores were busy for 24 % of a launch (profiles/r03_gconv32_f32_counters.txt).  Here a tile is 16 rows x 32
// channels on v_mfma_f32_16x16x4_f32 (the same exact-fp32 fma chain at the same rate, 512 cycles per active offset), a
// tile's offsets are fewer (the union is over 16 rows), and the waves of a block take the block's tiles off an LDS
// counter as they finish, so the matrix pipes of a CU stay fed until its tiles run out.  Results are those of the
// 32-row kernel up to the order of the 32 products inside one offset (channel order 8q + j instead of 16h + i); the
// order over the offsets is unchanged.
//   A: lane (r = lane & 15, q = lane >> 4) supplies row r, channels 8q .. 8q+7 (one contiguous 32-B read of its gathered row);
//      MFMA step j contracts channels {8q + j : q = 0..3}.
//   B: sW[k][cb][jq][q][n][e] = B[c = 8q + 4jq + e][col = 16cb + n]: one ds_read_b128 gives a lane four steps of a column block.
//   D: lane holds rows 4q .. 4q+3 of column n (+ 16 per column block).
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int SPLIT_GROUP = 2;    // the same for the three-piece form (its pieces and fragments want the registers)
constexpr int F16_GROUP = 3;      // offsets whose gathers a wave has in flight together (2 x 16 B per lane each)

// (body as a device function, like gconv32_bf16_body: sW = the filter image [K * 1024 floats] in LDS, sNextp = the
// block's tile counter, vbid / vgrid / nthreads = this product's grid)
//
// SPLIT (round 4): the same tiles on v_mfma_f32_16x16x32_bf16 with every fp32 number cut into three bf16 pieces (exact:
// split3_bf16) and the six leading piece products summed in fp32 -- 12 matrix instructions of 16 cycles per active offset
// instead of 16 of 32.  The A operand of that instruction is lane (r, q) <-> row r, channels 8q .. 8q+7: exactly the
// gathered registers, cut in place.  The filter pieces are B fragments sB[slot][cb][piece][lane] (16 B each, 6 KiB per
// offset): K - 1 offsets fill the LDS (26 x 6 KiB = 156 KiB at K = 27), the last offset's fragments stay in registers.
template <bool TRANSPOSE_W, int PK = 0, bool SPLIT = false>
__device__ __forceinline__ void gconv16_f32_body(float *sW, int *sNextp, int vbid, int vgrid, int nthreads,
                                                 const int *__restrict__ table, int mirror, int K, int identity_k,
                                                 long long R, const long long *__restrict__ r_dev,
                                                 const float *__restrict__ X, const float *__restrict__ W,
                                                 const float *__restrict__ bias, float *__restrict__ Y) {
    constexpr int GROUP = SPLIT ? SPLIT_GROUP : F16_GROUP;      // offsets in flight together
    int &sNext = *sNextp;
    if (threadIdx.x == 0) sNext = 0;
    uint4 *sB = reinterpret_cast<uint4 *>(sW);
    uint4 wc00, wc01, wc02, wc10, wc11, wc12;         // SPLIT: the pieces of offset K - 1, this lane's B fragments
    auto filter_frag = [&](int k, int cb, int l, uint4 &p1, uint4 &p2, uint4 &p3) {     // fragment of lane l = (n, q)
        const int n = l & 15, qq = l >> 4;
        f32x4 lo, hi;
        if (!TRANSPOSE_W) {
            const float *src = W + ((long long)k * 32 + 8 * qq) * 32 + 16 * cb + n;
            lo = f32x4{src[0], src[32], src[64], src[96]};
            hi = f32x4{src[128], src[160], src[192], src[224]};
        } else {
            const float *src = W + ((long long)k * 32 + 16 * cb + n) * 32 + 8 * qq;
            lo = *(const f32x4 *)src;
            hi = *(const f32x4 *)(src + 4);
        }
        split3_bf16x8(lo, hi, p1, p2, p3);
    };
    // SPLIT: offset kc keeps its fragments in registers -- the centre offset of a SubM layer (every tile uses it), the
    // middle one otherwise; the others sit in LDS slot k - (k > kc)
    const int kc = identity_k >= 0 ? identity_k : K / 2;
    if constexpr (SPLIT) {
        for (int f = threadIdx.x; f < (K - 1) * 128; f += nthreads) {
            const int slot = f >> 7, cb = (f >> 6) & 1, l = f & 63;
            uint4 p1, p2, p3;
            filter_frag(slot + (slot >= kc ? 1 : 0), cb, l, p1, p2, p3);
            uint4 *dst = sB + ((slot * 2 + cb) * 3) * 64 + l;
            dst[0] = p1;
            dst[64] = p2;
            dst[128] = p3;
        }
        {
            uint4 p1, p2, p3;
            filter_frag(kc, 0, threadIdx.x & 63, p1, p2, p3);
            wc00 = p1, wc01 = p2, wc02 = p3;
            filter_frag(kc, 1, threadIdx.x & 63, p1, p2, p3);
            wc10 = p1, wc11 = p2, wc12 = p3;
        }
    } else if (!TRANSPOSE_W) {
        for (int blk = threadIdx.x; blk < K * 64; blk += nthreads) {
            const int k = blk >> 6, c4 = (blk >> 3) & 7, j4 = blk & 7;          // channels 4 c4 .., columns 4 j4 ..
            const float *src = W + ((long long)k * 32 + c4 * 4) * 32 + j4 * 4;
            const f32x4 r0 = *(const f32x4 *)(src), r1 = *(const f32x4 *)(src + 32);
            const f32x4 r2 = *(const f32x4 *)(src + 64), r3 = *(const f32x4 *)(src + 96);
            const int q = c4 >> 1, jq = c4 & 1, cb = j4 >> 2, n0 = (j4 & 3) * 4;
            float *dst = sW + (((((k * 2 + cb) * 2 + jq) * 4 + q) * 16) + n0) * 4;
            *(f32x4 *)(dst + 0) = f32x4{r0.x, r1.x, r2.x, r3.x};
            *(f32x4 *)(dst + 4) = f32x4{r0.y, r1.y, r2.y, r3.y};
            *(f32x4 *)(dst + 8) = f32x4{r0.z, r1.z, r2.z, r3.z};
            *(f32x4 *)(dst + 12) = f32x4{r0.w, r1.w, r2.w, r3.w};
        }
    } else {
        for (int e = threadIdx.x; e < K * 256; e += nthreads) {
            const int k = e >> 8, col = (e >> 3) & 31, c4 = e & 7;
            const f32x4 v = *(const f32x4 *)(W + ((long long)k * 32 + col) * 32 + c4 * 4);
            const int q = c4 >> 1, jq = c4 & 1, cb = col >> 4, n = col & 15;
            *(f32x4 *)(sW + (((((k * 2 + cb) * 2 + jq) * 4 + q) * 16) + n) * 4) = v;
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    // XCD-aware: blocks with equal blockIdx % 8 share an L2 and take one contiguous range of the VALID tiles; inside it
    // block bi owns tiles bi, bi + bpx, ... (consecutive tiles -- one event's, alike in cost -- go to different CUs)
    const int xcd = vbid & 7, bi = vbid >> 3, bpx = vgrid >> 3;
    const long long Rv = valid_rows(R, r_dev);
    const long long nt_v = (Rv + 15) >> 4, tpx_v = (nt_v + 7) >> 3;
    const long long t_begin = (long long)xcd * tpx_v;
    const long long t_end = t_begin + tpx_v < nt_v ? t_begin + tpx_v : nt_v;
    const float bj0 = bias ? bias[r] : 0.f, bj1 = bias ? bias[16 + r] : 0.f;
    // the gathered rows through a raw buffer (byte offsets below 2 GiB: the dispatcher checks the row count)
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, 0x7FFFFFFF, 0x00020000);
    while (true) {
        int mine = 0;
        if (lane == 0) mine = atomicAdd(&sNext, 1);
        mine = __builtin_amdgcn_readfirstlane(mine);
        const long long tile = t_begin + bi + (long long)mine * bpx;
        if (tile >= t_end || tile * 16 >= Rv) break;
        const long long row = tile * 16 + r;
        const bool live = row < Rv;
        const long long rowc = live ? row : 0;
        // ---- phase 1: which offsets does the tile use?  Quarter q of the wave reads the entries of offsets q, q + 4, ...
        // (7 loads per lane instead of 27), all issued together; one ballot per offset over its quarter's lanes.
        int v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 4 * i + q;
            const int kk = k < K ? k : K - 1;
            if constexpr (PK == 0)
                v[i] = table[(long long)(mirror ? K - 1 - kk : kk) * R + rowc];
            else
                v[i] = packed_entry(table[(long long)(kk / PK) * R + rowc], kk % PK);       // packed table: K / PK rows
        }
        unsigned mask = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 4 * i + q;
            const bool ok = live && k < K && (k == identity_k || v[i] >= 0);
            const unsigned long long bal = __ballot(ok);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
                if (4 * i + qq < 32 && ((bal >> (16 * qq)) & 0xFFFFull) != 0ull) mask |= 1u << (4 * i + qq);
        }
        mask = __builtin_amdgcn_readfirstlane(mask);
        f32x4v acc0, acc1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc0[i] = bj0;
            acc1[i] = bj1;
        }
        if (mask != 0) {
            // ---- phase 2: groups of GROUP active offsets in ascending order: all the group's gathers are in flight
            // before its MFMAs.  The table entry of (row r, offset k) sits in register k >> 2 of lane (r, k & 3) since
            // phase 1: one select chain + one lane exchange per offset instead of a second read of the table.
            auto entry_of = [&](int k) -> int {
                const int i = k >> 2;
                int val = v[0];
#pragma unroll
                for (int t = 1; t < 8; ++t) val = (i == t) ? v[t] : val;
                int got = __shfl(val, r + 16 * (k & 3), 64);
                got = (k == identity_k) ? (int)rowc : got;
                return live ? got : -1;
            };
            // Software pipeline over the groups: while a group's MFMAs run, the NEXT group's gathers are in flight (two
            // register sets, the loop body written out twice so that they swap without moves).
            struct Group {
                int k[GROUP], nb[GROUP];
                f32x4 a[GROUP][2];
            };
            auto fetch = [&](Group &gr) {                     // takes the next GROUP offsets off the mask
#pragma unroll
                for (int g = 0; g < GROUP; ++g) {
                    gr.k[g] = mask ? __builtin_ctz(mask) : -1;
                    mask = mask ? (mask & (mask - 1)) : 0u;
                }
#pragma unroll
                for (int g = 0; g < GROUP; ++g) {
                    const int e = entry_of(gr.k[g] >= 0 ? gr.k[g] : 0);
                    gr.nb[g] = gr.k[g] >= 0 ? e : -1;
                }
#pragma unroll
                for (int g = 0; g < GROUP; ++g) {
                    // unconditional (loads behind branches make hipcc wait for all of them), through a raw buffer: a
                    // missing neighbour / empty slot points at or past the end of the buffer and reads as 0 -- no select
                    // on the 8 registers afterwards, a 32-bit offset instead of a 64-bit address
                    int voff = gr.nb[g] >= 0 ? (int)((unsigned)gr.nb[g] * 128u + (unsigned)q * 32u) : (int)0x80000000;
                    asm volatile("" : "+v"(voff));         // opaque: "+ 16" folds into the instruction's immediate
                    gr.a[g][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcX, voff, 0, 0));
                    gr.a[g][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcX, voff + 16, 0, 0));
                }
            };
            // SPLIT: one offset = cut the gathered registers, six fragments, twelve matrix instructions (smallest
            // products first; the two column blocks' chains alternate).  Written as a function of the six fragments and
            // called once per source: a select between register and LDS fragments would go through scratch.
#define WFS_SPLIT2(xa, pb0, pb1)                                                                                       \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8e, xa), __builtin_bit_cast(bf16x8e, pb0),   \
                                                   acc0, 0, 0, 0);                                                     \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8e, xa), __builtin_bit_cast(bf16x8e, pb1),   \
                                                   acc1, 0, 0, 0);
            auto six = [&](const uint4 &x1, const uint4 &x2, const uint4 &x3, const uint4 &b00, const uint4 &b01,
                           const uint4 &b02, const uint4 &b10, const uint4 &b11, const uint4 &b12) {
                WFS_SPLIT2(x3, b00, b10)
                WFS_SPLIT2(x1, b02, b12)
                WFS_SPLIT2(x2, b01, b11)
                WFS_SPLIT2(x2, b00, b10)
                WFS_SPLIT2(x1, b01, b11)
                WFS_SPLIT2(x1, b00, b10)
            };
#undef WFS_SPLIT2
            auto split_one = [&](const f32x4 &a0, const f32x4 &a1, int k) {          // k != kc: fragments from LDS
                uint4 x1, x2, x3;
                split3_bf16x8(a0, a1, x1, x2, x3);
                const uint4 *bp = sB + (k - (k > kc ? 1 : 0)) * 384 + lane;
                six(x1, x2, x3, bp[0], bp[64], bp[128], bp[192], bp[256], bp[320]);
            };
            auto multiply = [&](const Group &gr) {
                if constexpr (SPLIT) {
                    if (gr.k[GROUP - 1] >= 0) {
                        // a full group (every group but a tile's last): one basic block, so that the cutting and the
                        // fragment reads of an offset can be scheduled under the matrix instructions of its neighbour
#pragma unroll
                        for (int g = 0; g < GROUP; ++g) split_one(gr.a[g][0], gr.a[g][1], gr.k[g]);
                        return;
                    }
                }
#pragma unroll
                for (int g = 0; g < GROUP; ++g)
                    if (gr.k[g] >= 0) {
                        const f32x4 a0 = gr.a[g][0], a1 = gr.a[g][1];
                        if constexpr (SPLIT) {
                            split_one(a0, a1, gr.k[g]);
                            continue;
                        }
                        const f32x4 *bp = (const f32x4 *)(sW + ((gr.k[g] * 16 + q) * 16 + r) * 4);   // (k, cb 0, jq 0, q, n)
                        const f32x4 b00 = bp[0], b01 = bp[64], b10 = bp[128], b11 = bp[192];   // [cb][jq]: +64 f32x4 per jq, +128 per cb
                        // the two column blocks' chains alternate: a dependent MFMA never follows its producer directly
#define WFS_MFMA2(a, e, b0, b1)                                                       \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.e, b0.e, acc0, 0, 0, 0);            \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.e, b1.e, acc1, 0, 0, 0);
                        WFS_MFMA2(a0, x, b00, b10)
                        WFS_MFMA2(a0, y, b00, b10)
                        WFS_MFMA2(a0, z, b00, b10)
                        WFS_MFMA2(a0, w, b00, b10)
                        WFS_MFMA2(a1, x, b01, b11)
                        WFS_MFMA2(a1, y, b01, b11)
                        WFS_MFMA2(a1, z, b01, b11)
                        WFS_MFMA2(a1, w, b01, b11)
#undef WFS_MFMA2
                    }
            };
            Group ga, gb;
#pragma unroll
            for (int g = 0; g < GROUP; ++g) ga.k[g] = -1;
            if constexpr (SPLIT) {
                // the register-resident offset first: its rows are asked for together with the first group's and
                // multiplied while those are in flight
                const bool use_c = (mask >> kc) & 1u;
                mask &= ~(1u << kc);
                if (use_c) {
                    const int nbc = entry_of(kc);
                    int voff = nbc >= 0 ? (int)((unsigned)nbc * 128u + (unsigned)q * 32u) : (int)0x80000000;
                    asm volatile("" : "+v"(voff));
                    const f32x4 c0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcX, voff, 0, 0));
                    const f32x4 c1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcX, voff + 16, 0, 0));
                    if (mask != 0) fetch(ga);
                    uint4 x1, x2, x3;
                    split3_bf16x8(c0, c1, x1, x2, x3);
                    six(x1, x2, x3, wc00, wc01, wc02, wc10, wc11, wc12);
                } else {
                    fetch(ga);
                }
            } else {
                fetch(ga);
            }
            while (SPLIT ? ga.k[0] >= 0 : true) {
                const bool more_b = mask != 0;
                if (more_b) fetch(gb);
                __builtin_amdgcn_sched_barrier(0);
                multiply(ga);
                __builtin_amdgcn_sched_barrier(0);
                if (!more_b) break;
                const bool more_a = mask != 0;
                if (more_a) fetch(ga);
                __builtin_amdgcn_sched_barrier(0);
                multiply(gb);
                __builtin_amdgcn_sched_barrier(0);
                if (!more_a) break;
            }
        }
        // D map of the 16x16 MFMA: col = lane & 15 (+ 16 per column block), row = 4 (lane >> 4) + reg
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long orow = tile * 16 + 4 * q + i;
            if (orow < Rv) {
                Y[orow * 32 + r] = acc0[i];
                Y[orow * 32 + 16 + r] = acc1[i];
            }
        }
    }
}

template <bool TRANSPOSE_W, int PK = 0>
__global__ void __launch_bounds__(768) k_gconv16_split(const int *__restrict__ table, int mirror, int K, int identity_k,
                                                       long long R, const long long *__restrict__ r_dev,
                                                       const float *__restrict__ X, const float *__restrict__ W,
                                                       const float *__restrict__ bias, float *__restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) float sW_dyn[];
    __shared__ int sNext;
    gconv16_f32_body<TRANSPOSE_W, PK, true>(sW_dyn, &sNext, (int)blockIdx.x, (int)gridDim.x, (int)blockDim.x, table, mirror,
                                            K, identity_k, R, r_dev, X, W, bias, Y);
}

template <bool TRANSPOSE_W, int PK = 0>
__global__ void __launch_bounds__(1024) k_gconv16_f32(const int *__restrict__ table, int mirror, int K, int identity_k,
                                                      long long R, const long long *__restrict__ r_dev,
                                                      const float *__restrict__ X, const float *__restrict__ W,
                                                      const float *__restrict__ bias, float *__restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) float sW_dyn[];
    __shared__ int sNext;
    gconv16_f32_body<TRANSPOSE_W, PK>(sW_dyn, &sNext, (int)blockIdx.x, (int)gridDim.x, (int)blockDim.x, table, mirror, K,
                                      identity_k, R, r_dev, X, W, bias, Y);
}

// ------------------------------------------------------------------------------------------ 32 -> 32, bf16
// bf16 rows in HBM (64 B per voxel), fp32 master filters converted to bf16 while they are staged into LDS,
// fp32 accumulation: v_mfma_f32_32x32x16_bf16, 2 MFMAs per kernel offset and 32-row tile.  At 1/16 of the fp32
// MFMA cost the kernel is bound by memory latency, so it is organised for memory-level parallelism:
//   phase 1  the tile's K table entries are read once (coalesced) and parked in LDS; ballots give the mask of
//            offsets the tile uses;
//   phase 2  active offsets are taken GROUP (8) at a time: all their row gathers are issued back to back
//            (2 x 16 B per lane and offset), then the 16 MFMAs run while the next group's gathers are in flight.
// Channel order inside the contraction: lane (r, h), k-step s, element j <-> channel 16h + 8s + j, i.e. one
// contiguous 32-B read per lane and gathered row; the LDS filter image is built to match:
//   sWb[k][s][h][col][j] = B[16h + 8s + j][col],  B[c][col] = W[k][c][col] (forward) or W[k][col][c] (dX).
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BF_GROUP = 6;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// the kernels below are written once for both 16-bit row types H (wfs_bf16 / wfs_f16): rows travel as raw dwords,
// only the float <-> H conversions (wfs_pack2<H>, wfs_round_to<H>) and the MFMA opcode differ
template <typename H, typename V>
__device__ __forceinline__ f32x16 mfma16(V a, V b, f32x16 acc) {
    static_assert(sizeof(V) == 16, "8 x 16-bit operands");
    if constexpr (sizeof(H) == 2 && __is_same(H, wfs_f16))
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

// Timing knock-outs (tools/exp/knock.sh, profiles/r01_gconv32_bf16_knockouts.txt): -DWFS_KNOCK=bits builds a library
// whose k_gconv32_bf16 skips a phase -- 1 filter staging, 2 table reads, 4 gathers + MFMA, 8 stores, 16 gathers read the
// tile's own rows, 32 only one of a plane's three time offsets is gathered, 64 the dW body's gathers read the tile's
// own rows -- to measure what
// each phase costs inside a replayed graph.  Results are wrong by construction; 0 (the default) compiles to nothing.
#ifndef WFS_KNOCK
#define WFS_KNOCK 0
#endif
// (the body is a device function so that the backward pass can run it beside the dW kernel's body in ONE launch:
// k_bwd32_bf16 below; vbid / vgrid = this product's block index and block count, nthreads = its threads)
template <typename H, bool TRANSPOSE_W, int PK = 0>
__device__ __forceinline__ void gconv32_bf16_body(unsigned char *smem, int vbid, int vgrid, int nthreads,
                                                  const int *__restrict__ table, int mirror, int K, int identity_k,
                                                  long long R, const long long *__restrict__ r_dev,
                                                  const H *__restrict__ X, const float *__restrict__ W,
                                                  const float *__restrict__ bias, H *__restrict__ Y, long long ntiles,
                                                  long long tiles_per_xcd) {
    uint4 *sWb = reinterpret_cast<uint4 *>(smem);                           // K * 128 fragments of 16 B
    int *sNb = reinterpret_cast<int *>(smem + (size_t)K * 2048);            // [waves][K][32]
    // filter staging, WSB fragments per thread at a time: all their loads are issued (unconditionally, clamped)
    // before the first conversion, so a block pays one memory round trip per batch instead of one per fragment
    constexpr int WSB = 5;
    const int nfrag = K * 128;
    if (!(WFS_KNOCK & 1))
    for (int u0 = threadIdx.x; u0 < nfrag; u0 += nthreads * WSB) {
        float w[WSB][8];
#pragma unroll
        for (int b = 0; b < WSB; ++b) {
            int u = u0 + b * nthreads;
            u = u < nfrag ? u : nfrag - 1;
            int k = u >> 7, s = (u >> 6) & 1, h = (u >> 5) & 1, col = u & 31;
            int c0 = 16 * h + 8 * s;
            if (!TRANSPOSE_W) {
#pragma unroll
                for (int j = 0; j < 8; ++j) w[b][j] = W[((long long)k * 32 + c0 + j) * 32 + col];
            } else {
                const f32x4 *src = (const f32x4 *)(W + ((long long)k * 32 + col) * 32 + c0);
                f32x4 lo = src[0], hi = src[1];
                w[b][0] = lo.x; w[b][1] = lo.y; w[b][2] = lo.z; w[b][3] = lo.w;
                w[b][4] = hi.x; w[b][5] = hi.y; w[b][6] = hi.z; w[b][7] = hi.w;
            }
        }
#pragma unroll
        for (int b = 0; b < WSB; ++b) {
            int u = u0 + b * nthreads;
            uint4 v;
            v.x = wfs_pack2<H>(w[b][0], w[b][1]);
            v.y = wfs_pack2<H>(w[b][2], w[b][3]);
            v.z = wfs_pack2<H>(w[b][4], w[b][5]);
            v.w = wfs_pack2<H>(w[b][6], w[b][7]);
            if (u < nfrag) sWb[u] = v;
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = nthreads >> 6;
    const int r = lane & 31, h = lane >> 5;
    int *myNb = sNb + (size_t)wid * K * 32;
    // tile ranges per XCD cut from the VALID tiles (see k_gconv32_f32): the padding of a captured step costs nothing
    const int xcd = vbid & 7, bi = vbid >> 3, bpx = vgrid >> 3;
    const long long Rv = valid_rows(R, r_dev);
    const long long nt_v = (Rv + 31) >> 5, tpx_v = (nt_v + 7) >> 3;
    (void)ntiles;
    (void)tiles_per_xcd;
    const long long t_begin = (long long)xcd * tpx_v;
    const long long t_end = t_begin + tpx_v < nt_v ? t_begin + tpx_v : nt_v;
    const float bj = bias ? bias[r] : 0.f;
    // the gathered rows through a raw buffer (byte offsets below 2 GiB: the dispatcher checks the row count)
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, 0x7FFFFFFF, 0x00020000);
    for (long long tile = t_begin + (long long)wid * bpx + bi; tile < t_end; tile += (long long)bpx * nw) {
        if (tile * 32 >= Rv) break;
        const long long row = tile * 32 + r;
        const bool live = row < Rv;
        const long long rowc = live ? row : 0;
        // ---- phase 1
        int v[32];
        if constexpr (PK == 0) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                int kk = k < K ? k : K - 1;
                v[k] = (WFS_KNOCK & 2) ? -1 : table[(long long)(mirror ? K - 1 - kk : kk) * R + rowc];
            }
        } else {
            // packed table: K / PK loads instead of K (9 instead of 27 at the PSD geometry)
            constexpr int NQ = 32 / PK;
            int ev[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int qq = q * PK < K ? q : (K - 1) / PK;
                ev[q] = (WFS_KNOCK & 2) ? -1 : table[(long long)qq * R + rowc];
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = packed_entry(ev[k / PK < NQ ? k / PK : NQ - 1], k % PK);
        }
        unsigned mask = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            int nb = (k == identity_k) ? (int)rowc : v[k];
            nb = (live && k < K) ? nb : -1;
            if (k < K && h == 0) myNb[k * 32 + r] = nb;
            if (__ballot(nb >= 0) != 0ull) mask |= 1u << k;
        }
        mask = __builtin_amdgcn_readfirstlane(mask);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = bj;
        // ---- phase 2 (the wave's own LDS writes above are visible to its later reads: same wave, in order)
        if (WFS_KNOCK & 4) mask = 0;
        while (mask != 0) {
            int ks[BF_GROUP];
            uint4 a_lo[BF_GROUP], a_hi[BF_GROUP];
            int nbs[BF_GROUP];
#pragma unroll
            for (int g = 0; g < BF_GROUP; ++g) {
                ks[g] = mask ? __builtin_ctz(mask) : -1;
                mask = mask ? (mask & (mask - 1)) : 0u;
            }
#pragma unroll
            for (int g = 0; g < BF_GROUP; ++g)
                if (ks[g] >= 0) {
                    nbs[g] = myNb[ks[g] * 32 + r];
                    // WFS_KNOCK & 16: every offset reads the tile's OWN rows (a coalesced window instead of a gather):
                    // an upper bound on what run-structured window loads could save
                    // through a raw buffer: a missing neighbour points past the end of the buffer and reads as 0 (no select
                    // on the 8 registers afterwards), a 32-bit byte offset instead of a 64-bit address
                    const int src = (WFS_KNOCK & 16) ? (int)rowc : nbs[g];
                    int voff = src >= 0 ? (int)((unsigned)src * 64u + (unsigned)h * 32u) : (int)0x80000000;
                    asm volatile("" : "+v"(voff));
                    // WFS_KNOCK & 32: only the MIDDLE time offset of every (kx, ky) plane is gathered, its neighbours reuse
                    // a constant: the gather VOLUME of a kernel that stages one row window per plane in LDS (~34 rows
                    // instead of 3 x 32), without charging it anything for the staging -- an upper bound on that design
                    if ((WFS_KNOCK & 32) && (ks[g] % 3) != 1) {
                        a_lo[g] = uint4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
                        a_hi[g] = a_lo[g];
                    } else {
                        a_lo[g] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrcX, voff, 0, 0));
                        a_hi[g] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrcX, voff + 16, 0, 0));
                    }
                }
#pragma unroll
            for (int g = 0; g < BF_GROUP; ++g)
                if (ks[g] >= 0) {
                    const uint4 lo = a_lo[g], hi = a_hi[g];
                    const uint4 *bp = sWb + (size_t)ks[g] * 128 + h * 32 + r;
                    uint4 b0 = bp[0], b1 = bp[64];
                    acc = mfma16<H>(lo, b0, acc);
                    acc = mfma16<H>(hi, b1, acc);
                }
        }
        // ---- epilogue: reg i holds (row (i&3) + 8(i>>2) + 4h, col r).  Neighbouring columns are paired with one
        // lane exchange so that every lane stores one packed dword: even lanes row(i), odd lanes row(i+1).
        unsigned *Yw = reinterpret_cast<unsigned *>(Y);
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            float mine0 = acc[i], mine1 = acc[i + 1];
            float send = (lane & 1) ? mine0 : mine1;
            float got = __shfl_xor(send, 1, 64);
            unsigned packed = (lane & 1) ? wfs_pack2<H>(got, mine1) : wfs_pack2<H>(mine0, got);
            int ri = (lane & 1) ? i + 1 : i;
            long long orow = tile * 32 + (ri & 3) + 8 * (ri >> 2) + 4 * h;
            if (!(WFS_KNOCK & 8) || packed == 0x12345678u)
                if (orow < Rv) Yw[orow * 16 + (r >> 1)] = packed;
        }
    }
}

template <typename H, bool TRANSPOSE_W, int PK = 0>
__global__ void __launch_bounds__(1024) k_gconv32_bf16(const int *__restrict__ table, int mirror, int K,
                                                       int identity_k,
                                                       long long R, const long long *__restrict__ r_dev,
                                                       const H *__restrict__ X,
                                                       const float *__restrict__ W, const float *__restrict__ bias,
                                                       H *__restrict__ Y, long long ntiles,
                                                       long long tiles_per_xcd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    gconv32_bf16_body<H, TRANSPOSE_W, PK>(smem, (int)blockIdx.x, (int)gridDim.x, (int)blockDim.x, table, mirror, K,
                                          identity_k, R, r_dev, X, W, bias, Y, ntiles, tiles_per_xcd);
}

// ------------------------------------------------------------------------------------------ 2 -> 32
template <typename T>
__global__ void __launch_bounds__(256) k_gconv_c2c32(const int *__restrict__ table, KMap kmap, int K, int identity_k,
                                                     long long R, const long long *__restrict__ r_dev,
                                                     const T *__restrict__ X,
                                                     const float *__restrict__ W, const float *__restrict__ bias,
                                                     T *__restrict__ Y) {
    __shared__ __attribute__((aligned(16))) float sW[128 * 64];
    for (int e = threadIdx.x; e < K * 64; e += 256) sW[e] = W[e];
    __syncthreads();
    const int rsub = threadIdx.x >> 3, cq = threadIdx.x & 7;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *(const f32x4 *)(bias + cq * 4);
    const long long Rv = valid_rows(R, r_dev);
    for (long long row = (long long)blockIdx.x * 32 + rsub; row < Rv; row += (long long)gridDim.x * 32) {
        f32x4 acc = bv;
        for (int k = 0; k < K; ++k) {
            int nb = (k == identity_k) ? (int)row : table[(long long)kmap.v[k] * R + row];
            if (nb < 0) continue;
            float x0 = wfs_ld(X + (long long)nb * 2), x1 = wfs_ld(X + (long long)nb * 2 + 1);
            f32x4 w0 = *(const f32x4 *)(sW + k * 64 + cq * 4);
            f32x4 w1 = *(const f32x4 *)(sW + k * 64 + 32 + cq * 4);
            acc.x = fmaf(x0, w0.x, acc.x);
            acc.y = fmaf(x0, w0.y, acc.y);
            acc.z = fmaf(x0, w0.z, acc.z);
            acc.w = fmaf(x0, w0.w, acc.w);
            acc.x = fmaf(x1, w1.x, acc.x);
            acc.y = fmaf(x1, w1.y, acc.y);
            acc.z = fmaf(x1, w1.z, acc.z);
            acc.w = fmaf(x1, w1.w, acc.w);
        }
        T *y = Y + row * 32 + cq * 4;
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<f32x4 *>(y) = acc;
        } else {
            uint2 pk;
            pk.x = wfs_pack2<T>(acc.x, acc.y);
            pk.y = wfs_pack2<T>(acc.z, acc.w);
            *reinterpret_cast<uint2 *>(y) = pk;
        }
    }
}

// ------------------------------------------------------------------------------------------ dW 32 x 32
// dW[k][a][b] = sum_r S[r][a] * G[table[k][r]][b].   MFMA: D[i = a][j = b] += A[i][kk] B[kk][j] with the
// tile's rows as kk:  A[a][kk = 2s+h] = S[row0 + 2s + h][a],  B[kk][b] = G[nb(k, row0 + 2s + h)][b].
// Both operands are whole 128-B rows per half-wave -> perfectly coalesced loads.
// Work unit = (block of rows, group of DW_KG offsets {g, g+NG, g+2NG, ...}); the 8 waves of a block take
// interleaved tiles of the block's row range and are summed through LDS; slabs are reduced afterwards.
constexpr int DW_KG = 4;        // offsets per wave (4 x 16 accumulator registers)
constexpr int DW_WAVES = 8;

constexpr int DW_LDS = DW_WAVES * 1024 * 4;                        // block reduction staging, 32 KiB
// (body as a device function; bx / by / nbx = the block's coordinates in this product's grid)
template <typename T>
__device__ __forceinline__ void gdw32_body(float *sRed, int bx, int by, int nbx, const int *__restrict__ table, int pk, int K,
                                           int identity_k, long long Rcap, const long long *__restrict__ r_dev,
                                           const T *__restrict__ S, const T *__restrict__ G, float *__restrict__ part,
                                           int ngroups) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int g = by;
    const long long R = valid_rows(Rcap, r_dev);            // rows to process; Rcap stays the table stride
    const long long ntiles = (R + 31) >> 5;
    static_assert(sizeof(T) == 4, "fp32 rows");
    // raw buffers (stride 0, byte offsets): S holds R valid rows of 128 B; G is addressed below 2 GiB, anything at or
    // above reads as 0 (the launcher checks both sizes)
    const __amdgpu_buffer_rsrc_t rsrcS = __builtin_amdgcn_make_buffer_rsrc((void *)S, 0, (int)(R * 128), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcG = __builtin_amdgcn_make_buffer_rsrc((void *)G, 0, 0x7FFFFFFF, 0x00020000);
    const unsigned j4 = (unsigned)j * 4u;
    // consecutive tiles go to different blocks: an event's tiles (similar numbers of active offsets) spread over the chip
    f32x16 acc[DW_KG];
#pragma unroll
    for (int q = 0; q < DW_KG; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
    int trw[DW_KG], osel[DW_KG];
#pragma unroll
    for (int q = 0; q < DW_KG; ++q) {
        const int k = g + q * ngroups;
        table_row_of(pk, k < K ? k : K - 1, &trw[q], &osel[q]);
    }
    for (long long tile = bx + (long long)wid * nbx; tile < ntiles; tile += (long long)DW_WAVES * nbx) {
        const long long row0 = tile * 32;
        // lane j holds the table entries of row (row0 + j) for this wave's offsets.  Loads are unconditional
        // on clamped addresses (see k_gconv32_f32), validity is applied afterwards.
        const long long trow = row0 + j < R ? row0 + j : R - 1;
        int nbv[DW_KG];
#pragma unroll
        for (int q = 0; q < DW_KG; ++q) nbv[q] = table_value(table[(long long)trw[q] * Rcap + trow], osel[q]);
        // S rows through a raw buffer whose size is the VALID rows: one lane offset per tile, the 16 rows of a lane by
        // the instruction's immediate offset, rows past the end read as 0 -- no address arithmetic, no select per load
        // (the pointer form spent ~11 VALU instructions on each of a tile's 16 + 16 x active offsets loads: 16 us of VALU
        // per SIMD against 11 us of MFMA, profiles/r03_gdw32_f32_counters.txt)
        float a[16];
        {
            int voff = (int)((unsigned)(row0 + h) * 128u + (unsigned)j * 4u);
            asm volatile("" : "+v"(voff));         // opaque: keeps "+ s * 256" a constant that folds into the immediate
#pragma unroll
            for (int s = 0; s < 16; ++s)
                a[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrcS, voff + s * 256, 0, 0));
        }
        bool any = false;
#pragma unroll
        for (int q = 0; q < DW_KG; ++q) {
            int k = g + q * ngroups;
            int nb = (k == identity_k) ? (int)trow : nbv[q];
            nb = (k < K && row0 + j < R) ? nb : -1;
            nbv[q] = nb;
            any = any || (__ballot(nb >= 0) != 0ull);
        }
        if (!any) continue;
        // one ACTIVE offset at a time (inactive ones -- most of a SubM tile's -- issue nothing): 16 registers of gathered
        // rows instead of 64 leave room for two blocks = four waves per SIMD on a CU, whose MFMAs and address arithmetic
        // run under this wave's gather round trip
#pragma unroll
        for (int q = 0; q < DW_KG; ++q) {
            if (__ballot(nbv[q] >= 0) == 0ull) continue;
            float b[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                // the two rows of an MFMA step are wave-uniform: their byte offsets are scalar work, a missing row gets an
                // offset past the end of the buffer and reads as 0
                const int n0 = __builtin_amdgcn_readlane(nbv[q], 2 * s);
                const int n1 = __builtin_amdgcn_readlane(nbv[q], 2 * s + 1);
                const unsigned o0 = n0 >= 0 ? (unsigned)n0 * 128u : 0x80000000u;
                const unsigned o1 = n1 >= 0 ? (unsigned)n1 * 128u : 0x80000000u;
                b[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrcG, (h ? o1 : o0) + j4, 0, 0));
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc[q], 0, 0, 0);
        }
    }
    // deterministic block reduction, one offset at a time: the 8 waves park that offset's accumulator in LDS, then
    // every thread adds two output elements over the waves in wave order
#pragma unroll
    for (int q = 0; q < DW_KG; ++q) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int arow = (i & 3) + 8 * (i >> 2) + 4 * h;
            sRed[wid * 1024 + arow * 32 + j] = acc[q][i];
        }
        __syncthreads();
        const int k = g + q * ngroups;
        for (int e = threadIdx.x; e < 1024; e += 512) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < DW_WAVES; ++w) v += sRed[w * 1024 + e];
            if (k < K) part[((long long)bx * K + k) * 1024 + e] = v;
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(512, 4) k_gdw32(const int *__restrict__ table, int pk, int K, int identity_k, long long Rcap,
                                                  const long long *__restrict__ r_dev, const T *__restrict__ S,
                                                  const T *__restrict__ G,
                                                  float *__restrict__ part, int ngroups, long long tiles_per_block) {
    __shared__ float sRed[DW_WAVES * 1024];
    (void)tiles_per_block;                                   // cut from the capacity: the valid tiles are shared out instead
    gdw32_body<T>(sRed, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, table, pk, K, identity_k, Rcap, r_dev, S, G, part,
                  ngroups);
}

// ------------------------------------------------------------------------------------------ 2 -> 32, fp32 MFMA
// First layer in exact fp32 on the matrix cores: Y[row][co] = b[co] + sum_k sum_ch X[nb(k,row)][ch] W[k][ch][co] is, per
// kernel offset, ONE v_mfma_f32_32x32x2_f32 on the 32-row tile (contraction = the 2 input channels): lane (r, h) supplies
// A[row r][ch h] = the gathered element and B[ch h][co r] = W[k][h][r], which it keeps in a register per offset.  One
// tile per wave: all K table entries of the lane's row are asked for together, then all K gathers (two memory round
// trips per tile; the VALU form walked its rows one after the other, two round trips EACH: 38 us -> see DESIGN.md 4).
__global__ void __launch_bounds__(256) k_gconv_c2c32_f32(const int *__restrict__ table, int mirror, int K, int identity_k,
                                                         long long R, const long long *__restrict__ r_dev,
                                                         const float *__restrict__ X, const float *__restrict__ W,
                                                         const float *__restrict__ bias, float *__restrict__ Y) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    float wreg[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) wreg[k] = k < K ? W[((long long)k * 2 + h) * 32 + r] : 0.f;
    const float bj = bias ? bias[r] : 0.f;
    const long long Rv = valid_rows(R, r_dev);
    const long long ntiles = (Rv + 31) >> 5;
    for (long long tile = (long long)blockIdx.x * nw + wid; tile < ntiles; tile += (long long)gridDim.x * nw) {
        const long long row = tile * 32 + r;
        const bool live = row < Rv;
        const long long rowc = live ? row : 0;
        int nb[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const int kk = k < K ? k : K - 1;
            nb[k] = table[(long long)(mirror ? K - 1 - kk : kk) * R + rowc];
        }
        float xv[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const int n = (k == identity_k) ? (int)rowc : nb[k];
            const bool ok = live && k < K && n >= 0;
            nb[k] = ok ? 1 : 0;
            xv[k] = X[(long long)(ok ? n : 0) * 2 + h];
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = bj;
#pragma unroll
        for (int k = 0; k < 27; ++k)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(nb[k] ? xv[k] : 0.f, wreg[k], acc, 0, 0, 0);
        // reg i holds (row (i & 3) + 8 (i >> 2) + 4 h, column r): 32 lanes store one 128-byte row
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long long orow = tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (orow < Rv) Y[orow * 32 + r] = acc[i];
        }
    }
}

// ------------------------------------------------------------------------------------------ 2 -> 32, bf16 MFMA
// First layer on the matrix cores: the K offsets x 2 input channels form a 64-wide contraction (54 used), so one
// 32-row tile is 4 x v_mfma_f32_32x32x16_bf16.  A-fragment of lane (r, h), k-step s = the 4 gathered dwords
// (2 bf16 channels each) of offsets 8s + 4h + {0,1,2,3} of row r -- read straight from global memory, no LDS, no
// transpose; all 16 table entries of a lane are loaded together, then all 16 gathers.  The filter image
// sWc[s][h][co][j] = W[k = 8s + 4h + j/2][c = j & 1][co] (zero for k >= K) is 4 KiB.
template <typename H>
__global__ void __launch_bounds__(256) k_gconv_c2c32_bf16(const int *__restrict__ table, int mirror, int K,
                                                         int identity_k, long long R,
                                                         const long long *__restrict__ r_dev,
                                                         const H *__restrict__ X, const float *__restrict__ W,
                                                         const float *__restrict__ bias, H *__restrict__ Y) {
    __shared__ __attribute__((aligned(16))) uint4 sWc[4 * 2 * 32];
    if (threadIdx.x < 256) {
        const int u = threadIdx.x;                 // 256 threads = 4 steps x 2 halves x 32 output channels
        const int st = u >> 6, hh = (u >> 5) & 1, co = u & 31;
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int k = 8 * st + 4 * hh + (j >> 1);
            w[j] = k < K ? W[((long long)k * 2 + (j & 1)) * 32 + co] : 0.f;
        }
        uint4 v;
        v.x = wfs_pack2<H>(w[0], w[1]);
        v.y = wfs_pack2<H>(w[2], w[3]);
        v.z = wfs_pack2<H>(w[4], w[5]);
        v.w = wfs_pack2<H>(w[6], w[7]);
        sWc[u] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long long Rv = valid_rows(R, r_dev);
    const long long ntiles = (Rv + 31) >> 5;
    const float bj = bias ? bias[r] : 0.f;
    const unsigned *Xw = reinterpret_cast<const unsigned *>(X);
    const int nw = blockDim.x >> 6;
    for (long long tile = (long long)blockIdx.x * nw + wid; tile < ntiles; tile += (long long)gridDim.x * nw) {
        const long long row = tile * 32 + r;
        const bool live = row < Rv;
        const long long rowc = live ? row : 0;
        int nb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            int k = 8 * (q >> 2) + 4 * h + (q & 3);
            int kk = k < K ? k : K - 1;
            nb[q] = table[(long long)(mirror ? K - 1 - kk : kk) * R + rowc];
        }
        unsigned xv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            int k = 8 * (q >> 2) + 4 * h + (q & 3);
            int n = (k == identity_k) ? (int)rowc : nb[q];
            bool ok = live && k < K && n >= 0;
            unsigned v = Xw[ok ? n : 0];
            xv[q] = ok ? v : 0u;
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = bj;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            uint4 a = {xv[4 * st], xv[4 * st + 1], xv[4 * st + 2], xv[4 * st + 3]};
            uint4 b = sWc[(st * 2 + h) * 32 + r];
            acc = mfma16<H>(a, b, acc);
        }
        unsigned *Yw = reinterpret_cast<unsigned *>(Y);
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            float mine0 = acc[i], mine1 = acc[i + 1];
            float send = (lane & 1) ? mine0 : mine1;
            float got = __shfl_xor(send, 1, 64);
            unsigned packed = (lane & 1) ? wfs_pack2<H>(got, mine1) : wfs_pack2<H>(mine0, got);
            int ri = (lane & 1) ? i + 1 : i;
            long long orow = tile * 32 + (ri & 3) + 8 * (ri >> 2) + 4 * h;
            if (orow < Rv) Yw[orow * 16 + (r >> 1)] = packed;
        }
    }
}

// ------------------------------------------------------------------------------------------ dW 32 x 32, bf16
// dW[k][a][b] = sum_r S[r][a] * G[table[k][r]][b] with bf16 rows and v_mfma_f32_32x32x16_bf16.  The contraction
// runs over ROWS, while memory holds a row's 32 channels contiguously, so both operands need a transpose: each
// wave parks the 32-row S tile and the gathered G tile in its own LDS region ([row][32 ch], written as whole
// 16-B chunks) and reads its fragments column-wise (lane (c, h), k-step s, element j <-> row 16s + 8h + j).
// Work unit: block = (row split, set of 4 offsets {g, g+7, g+14, g+21}); its 16 waves take interleaved tiles,
// skip (tile, offset) pairs without neighbours, and are summed through LDS in wave order (deterministic).
constexpr int DWB_WAVES = 16;
constexpr int DWB_KG = 4;

// The same fragment with gfx950's transposing LDS read: ds_read_b64_tr_b16 hands each 16-lane group a block of
// 4 rows x 16 columns column-major (lane 4q+p supplies the address of row q, columns 4p..4p+3; lane i receives column
// i of the 4 rows), so the lane (c = lane & 31, h = lane >> 5) gets rows row0 .. row0+7 of column c, row0 = 16 s + 8 h,
// from two reads instead of eight 16-bit reads and their packing.  Conflict-free on plain 64-byte rows (a 32-lane
// half covers 4 whole rows = 64 banks).  Needs EXEC all ones: call only from wave-uniform control flow.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 lds_column_frag_tr(const unsigned short *tile, int lane, int s) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const unsigned short *a = tile + (16 * s + 8 * (g >> 1) + q) * 32 + 16 * (g & 1) + 4 * p;
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)a);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + 4 * 32));
    s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 lds_column_frag(const unsigned short *tile, int col, int row0) {
    unsigned w[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        unsigned lo = tile[(row0 + 2 * m) * 32 + col];
        unsigned hi = tile[(row0 + 2 * m + 1) * 32 + col];
        w[m] = lo | (hi << 16);
    }
    uint4 v = {w[0], w[1], w[2], w[3]};
    return __builtin_bit_cast(bf16x8, v);
}

constexpr int DWB_LDS = DWB_WAVES * 2 * 32 * 32 * 2;          // per wave: S tile, G tile (64 KiB)
// (body as a device function: see gconv32_bf16_body; bx / by / nbx = the block's coordinates in this product's grid)
template <typename H>
__device__ __forceinline__ void gdw32_bf16_body(unsigned char *smem, int bx, int by, int nbx,
                                                const int *__restrict__ table, int pk, int K, int identity_k,
                                                long long Rcap, const long long *__restrict__ r_dev,
                                                const H *__restrict__ S, const H *__restrict__ G,
                                                float *__restrict__ part, int ngroups) {
    unsigned short(*sTiles)[2][32 * 32] = reinterpret_cast<unsigned short(*)[2][32 * 32]>(smem);   // [DWB_WAVES][2][1024]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;          // fragment coordinates
    const int grow = lane >> 2, gchunk = lane & 3;   // staging coordinates: rows grow and grow+16, 16-B chunk gchunk
    unsigned short *sS = sTiles[wid][0], *sG = sTiles[wid][1];
    const int g = by;
    const long long R = valid_rows(Rcap, r_dev);            // rows to process; Rcap stays the table stride
    const long long ntiles = (R + 31) >> 5;
    // cut from the VALID tiles (not the capacity): the padding of a captured step costs nothing
    const long long tpb = (ntiles + nbx - 1) / nbx;
    const long long t_begin = (long long)bx * tpb;
    const long long t_end = t_begin + tpb < ntiles ? t_begin + tpb : ntiles;
    f32x16 acc[DWB_KG];
#pragma unroll
    for (int q = 0; q < DWB_KG; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
    int trw[DWB_KG], osel[DWB_KG];
#pragma unroll
    for (int q = 0; q < DWB_KG; ++q) {
        const int k = g + q * ngroups;
        table_row_of(pk, k < K ? k : K - 1, &trw[q], &osel[q]);
    }
    for (long long tile = t_begin + wid; tile < t_end; tile += DWB_WAVES) {
        const long long row0 = tile * 32;
        const long long ra = row0 + grow, rb = row0 + grow + 16;
        const bool la = ra < R, lb = rb < R;
        const long long rac = la ? ra : R - 1, rbc = lb ? rb : R - 1;
        // table entries of the two rows this lane stages, for the block's 4 offsets (unconditional loads)
        int ta[DWB_KG], tb[DWB_KG];
#pragma unroll
        for (int q = 0; q < DWB_KG; ++q) {
            ta[q] = table_value(table[(long long)trw[q] * Rcap + rac], osel[q]);
            tb[q] = table_value(table[(long long)trw[q] * Rcap + rbc], osel[q]);
        }
        // the S tile is needed whenever any offset is active; issue its loads together with the table reads
        uint4 s0 = *(const uint4 *)(S + rac * 32 + gchunk * 8);
        uint4 s1 = *(const uint4 *)(S + rbc * 32 + gchunk * 8);
        unsigned long long act[DWB_KG];
        bool any = false;
#pragma unroll
        for (int q = 0; q < DWB_KG; ++q) {
            int k = g + q * ngroups;
            int na = (k == identity_k) ? (int)rac : ta[q], nb = (k == identity_k) ? (int)rbc : tb[q];
            ta[q] = (k < K && la) ? na : -1;
            tb[q] = (k < K && lb) ? nb : -1;
            act[q] = __ballot(ta[q] >= 0 || tb[q] >= 0);
            any = any || act[q] != 0ull;
        }
        if (!any) continue;
        *(uint4 *)(sS + grow * 32 + gchunk * 8) = keep_if(s0, la);
        *(uint4 *)(sS + (grow + 16) * 32 + gchunk * 8) = keep_if(s1, lb);
        __builtin_amdgcn_wave_barrier();
        bf16x8 a0, a1;
        // the gathers of ALL the block's offsets are issued together and unconditionally (clamped rows): one memory
        // round trip per tile; a load under the per-offset branch would cost one per offset (hipcc waits vmcnt(0))
        uint4 g0[DWB_KG], g1[DWB_KG];
#pragma unroll
        for (int q = 0; q < DWB_KG; ++q) {
            // WFS_KNOCK & 64: the gathers read the tile's OWN rows (what the fetch of a perfectly local table would be)
            g0[q] = *(const uint4 *)(G + (long long)(ta[q] >= 0 ? ((WFS_KNOCK & 64) ? rac : ta[q]) : 0) * 32 + gchunk * 8);
            g1[q] = *(const uint4 *)(G + (long long)(tb[q] >= 0 ? ((WFS_KNOCK & 64) ? rbc : tb[q]) : 0) * 32 + gchunk * 8);
        }
        a0 = lds_column_frag_tr(sS, lane, 0), a1 = lds_column_frag_tr(sS, lane, 1);
#pragma unroll
        for (int q = 0; q < DWB_KG; ++q) {
            if (act[q] == 0ull) continue;
            __builtin_amdgcn_wave_barrier();           // the previous offset's fragment reads are done (in order)
            *(uint4 *)(sG + grow * 32 + gchunk * 8) = keep_if(g0[q], ta[q] >= 0);
            *(uint4 *)(sG + (grow + 16) * 32 + gchunk * 8) = keep_if(g1[q], tb[q] >= 0);
            __builtin_amdgcn_wave_barrier();
            const bf16x8 b0 = lds_column_frag_tr(sG, lane, 0), b1 = lds_column_frag_tr(sG, lane, 1);
            acc[q] = mfma16<H>(a0, b0, acc[q]);
            acc[q] = mfma16<H>(a1, b1, acc[q]);
        }
    }
    // deterministic block reduction, one offset at a time: all 16 waves park that offset's accumulator in LDS (the
    // tile staging area is free now: 16 x 4 KiB), then every thread adds ONE output element over the waves in wave
    // order.  (Was: 16 serial read-modify-write passes over a shared 16 KiB accumulator, ~8 us of the kernel.)
    __syncthreads();
    float *sRed = reinterpret_cast<float *>(&sTiles[0][0][0]);          // [DWB_WAVES][1024]
#pragma unroll
    for (int q = 0; q < DWB_KG; ++q) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int arow = (i & 3) + 8 * (i >> 2) + 4 * h;
            sRed[wid * 1024 + arow * 32 + c] = acc[q][i];
        }
        __syncthreads();
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < DWB_WAVES; ++w) v += sRed[w * 1024 + threadIdx.x];
        const int k = g + q * ngroups;
        if (k < K) part[((long long)bx * K + k) * 1024 + threadIdx.x] = v;
        __syncthreads();
    }
}

template <typename H>
__global__ void __launch_bounds__(1024) k_gdw32_bf16(const int *__restrict__ table, int pk, int K, int identity_k,
                                                     long long Rcap, const long long *__restrict__ r_dev,
                                                     const H *__restrict__ S, const H *__restrict__ G,
                                                     float *__restrict__ part, int ngroups, long long tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    (void)tiles_per_block;
    gdw32_bf16_body<H>(smem, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, table, pk, K, identity_k, Rcap, r_dev, S, G,
                       part, ngroups);
}

// ------------------------------------------------------------------------------------------ dW 32 x 32, fp32 rows in three pieces
// fp32 rows on the bf16 matrix cores without giving up a bit (round 4).  A float's 24-bit significand cut 8 / 8 / 8 by
// truncation is three bf16 numbers whose sum is the float EXACTLY (p1 = the high half of the word, p2 = the high half of
// x - p1, p3 = x - p1 - p2: every subtraction is exact); products of two pieces are exact in fp32.  Of the nine piece
// products the six with orders (1,1) (1,2) (2,1) (2,2) (1,3) (3,1) are summed -- the three left out are below 2^-24 of
// the leading one, the size of one fp32 rounding -- by v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 12 matrix
// instructions of 32 cycles per (tile, offset) instead of k_gdw32<float>'s 16 v_mfma_f32_32x32x2_f32 of 64 cycles.
// The pieces are made where the tile is staged (fp32 rows in registers -> three [32][32] bf16 planes in the wave's LDS
// region), operands then read column-wise as in k_gdw32_bf16.  Work units, block reduction and slabs as k_gdw32<float>.
constexpr int DWS_WAVES = 12;
constexpr int DWS_KG = 4;
constexpr int DWS_LDS = DWS_WAVES * 6 * 2048;          // per wave: 3 planes of the S tile, 3 of the gathered tile (144 KiB)

// (body as a device function: bx / g / nbx = the block's coordinates in this product's grid)
__device__ __forceinline__ void gdw32_split_body(unsigned char *smem, int bx, int g, int nbx, const int *__restrict__ table,
                                                 int pk, int K, int identity_k, long long Rcap,
                                                 const long long *__restrict__ r_dev, const float *__restrict__ S,
                                                 const float *__restrict__ G, float *__restrict__ part, int ngroups) {
    unsigned short(*sTiles)[6][32 * 32] = reinterpret_cast<unsigned short(*)[6][32 * 32]>(smem);   // [DWS_WAVES][6][1024]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;          // fragment coordinates
    const int srow = lane >> 3, chunk = lane & 7;    // staging coordinates: rows srow + 8 p, 16-B chunk (4 floats)
    const long long R = valid_rows(Rcap, r_dev);
    const long long ntiles = (R + 31) >> 5;
    const __amdgpu_buffer_rsrc_t rsrcS = __builtin_amdgcn_make_buffer_rsrc((void *)S, 0, (int)(R * 128), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcG = __builtin_amdgcn_make_buffer_rsrc((void *)G, 0, 0x7FFFFFFF, 0x00020000);
    f32x16 acc[DWS_KG];
#pragma unroll
    for (int q = 0; q < DWS_KG; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
    int trw[DWS_KG], osel[DWS_KG];
#pragma unroll
    for (int q = 0; q < DWS_KG; ++q) {
        const int k = g + q * ngroups;
        table_row_of(pk, k < K ? k : K - 1, &trw[q], &osel[q]);
    }
    unsigned short *sS = sTiles[wid][0], *sG = sTiles[wid][3];
    const int soff = srow * 32 + chunk * 4;          // this lane's place in a plane (bf16 elements), + 256 per pass
    // consecutive tiles go to different blocks: an event's tiles (similar numbers of active offsets) spread over the chip.
    // (Asking for the NEXT tile's table entries and S rows under the current tile measured nothing here -- 35.6 vs 34.4
    // us -- and cost 20 register copies per tile.)
    const long long tstep = (long long)DWS_WAVES * nbx;
    for (long long tile = bx + (long long)wid * nbx; tile < ntiles; tile += tstep) {
        const long long row0 = tile * 32;
        const long long trow = row0 + c < R ? row0 + c : R - 1;
        int nbv[DWS_KG];
#pragma unroll
        for (int q = 0; q < DWS_KG; ++q) nbv[q] = table_value(table[(long long)trw[q] * Rcap + (trow > 0 ? trow : 0)], osel[q]);
        // the S tile through a raw buffer sized by the VALID rows (rows past the end read as 0)
        f32x4 sv[4];
        {
            int voff = (int)((unsigned)(row0 + srow) * 128u + (unsigned)chunk * 16u);
            asm volatile("" : "+v"(voff));
#pragma unroll
            for (int p = 0; p < 4; ++p)
                sv[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcS, voff + p * 1024, 0, 0));
        }
        unsigned long long act[DWS_KG];
        bool any = false;
#pragma unroll
        for (int q = 0; q < DWS_KG; ++q) {
            const int k = g + q * ngroups;
            int nb = (k == identity_k) ? (int)trow : nbv[q];
            nb = (k < K && row0 + c < R) ? nb : -1;
            nbv[q] = nb;
            act[q] = __ballot(nb >= 0);
            any = any || act[q] != 0ull;
        }
        if (any) {
        // gathers: two offsets in flight at a time (a missing row gets an offset past the end and reads as 0); the next
        // offset's rows are asked for as soon as an offset's registers are cut
        f32x4 gv[DWS_KG][4];
        auto issue = [&](int q) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int nb = __shfl(nbv[q], srow + 8 * p, 64);
                const unsigned off = nb >= 0 ? (unsigned)nb * 128u + (unsigned)chunk * 16u : 0x80000000u;
                gv[q][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcG, (int)off, 0, 0));
            }
        };
        issue(0);
        issue(1);
        __builtin_amdgcn_wave_barrier();               // the previous tile's fragment reads are done (LDS is in order)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint2 p1, p2, p3;
            split3_bf16(sv[p], true, p1, p2, p3);
            *(uint2 *)(sS + soff + p * 256) = p1;
            *(uint2 *)(sS + 1024 + soff + p * 256) = p2;
            *(uint2 *)(sS + 2048 + soff + p * 256) = p3;
        }
        __builtin_amdgcn_wave_barrier();
        bf16x8 a[3][2];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            a[pl][0] = lds_column_frag_tr(sS + pl * 1024, lane, 0);
            a[pl][1] = lds_column_frag_tr(sS + pl * 1024, lane, 1);
        }
#pragma unroll
        for (int q = 0; q < DWS_KG; ++q) {
            if (act[q] != 0ull) {
            __builtin_amdgcn_wave_barrier();           // the previous offset's fragment reads are done
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                uint2 p1, p2, p3;
                split3_bf16(gv[q][p], true, p1, p2, p3);
                *(uint2 *)(sG + soff + p * 256) = p1;
                *(uint2 *)(sG + 1024 + soff + p * 256) = p2;
                *(uint2 *)(sG + 2048 + soff + p * 256) = p3;
            }
            __builtin_amdgcn_wave_barrier();
            bf16x8 b[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                b[pl][0] = lds_column_frag_tr(sG + pl * 1024, lane, 0);
                b[pl][1] = lds_column_frag_tr(sG + pl * 1024, lane, 1);
            }
            // smallest products first: they meet in the accumulator before the leading one swamps them
#define WFS_SPLIT_MFMA(i, j)                                                                    \
    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[q], 0, 0, 0);         \
    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[q], 0, 0, 0);
            WFS_SPLIT_MFMA(2, 0)
            WFS_SPLIT_MFMA(0, 2)
            WFS_SPLIT_MFMA(1, 1)
            WFS_SPLIT_MFMA(1, 0)
            WFS_SPLIT_MFMA(0, 1)
            WFS_SPLIT_MFMA(0, 0)
#undef WFS_SPLIT_MFMA
            }
            if (q + 2 < DWS_KG) issue(q + 2);
        }
        }
    }
    // deterministic block reduction, one offset at a time (as k_gdw32): the waves park that offset's accumulator in LDS
    // (the tile planes are free now), every thread adds two output elements over the waves in wave order
    float *sRed = reinterpret_cast<float *>(smem);                      // [DWS_WAVES][1024]
#pragma unroll
    for (int q = 0; q < DWS_KG; ++q) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int arow = (i & 3) + 8 * (i >> 2) + 4 * h;
            sRed[wid * 1024 + arow * 32 + c] = acc[q][i];
        }
        __syncthreads();
        const int k = g + q * ngroups;
        for (int e = threadIdx.x; e < 1024; e += DWS_WAVES * 64) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < DWS_WAVES; ++w) v += sRed[w * 1024 + e];
            if (k < K) part[((long long)bx * K + k) * 1024 + e] = v;
        }
    }
}

__global__ void __launch_bounds__(DWS_WAVES * 64) k_gdw32_split(const int *__restrict__ table, int pk, int K, int identity_k,
                                                               long long Rcap, const long long *__restrict__ r_dev,
                                                               const float *__restrict__ S, const float *__restrict__ G,
                                                               float *__restrict__ part, int ngroups) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    gdw32_split_body(smem, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, table, pk, K, identity_k, Rcap, r_dev, S, G, part,
                     ngroups);
}

// dW and dX of an fp32 32 -> 32 layer in one launch, as k_bwd32_bf16 below: the three-piece bodies have the same block
// size (12 waves) and nearly the same LDS (144 / 156 KiB), which the fp32-instruction bodies did not
template <int PK>
__global__ void __launch_bounds__(DWS_WAVES * 64) k_bwd32_split(const int *__restrict__ table, int K, int identity_k,
                                                               long long Rcap, const long long *__restrict__ r_dev,
                                                               const float *__restrict__ S, const float *__restrict__ G,
                                                               const float *__restrict__ W, float *__restrict__ dX,
                                                               float *__restrict__ part, int ngroups, int nbx_dw,
                                                               int n_dw_pad, int n_dx, int dx_threads) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int sNext;
    const int bid = (int)blockIdx.x;
    if (bid < n_dw_pad) {
        if (bid >= nbx_dw * ngroups) return;
        gdw32_split_body(smem, bid % nbx_dw, bid / nbx_dw, nbx_dw, table, PK, K, identity_k, Rcap, r_dev, S, G, part, ngroups);
    } else {
        if ((int)threadIdx.x >= dx_threads) return;
        gconv16_f32_body<true, PK, true>(reinterpret_cast<float *>(smem), &sNext, bid - n_dw_pad, n_dx, dx_threads, table, 0,
                                         K, identity_k, Rcap, r_dev, G, W, nullptr, dX);
    }
}

// dW and dX of a 32 -> 32 layer in ONE launch (round 4): both gather dY through the same by-input table and neither
// reads what the other writes, but as two launches the second waits for the first one's last block and pays a kernel
// boundary of its own (~4.6 us inside a captured step, as much as its bytes).  Blocks [0, n_dw) run the dW body --
// grid (nbx_dw, ngroups), rounded up to a multiple of 8 blocks so that the dX blocks keep their XCD (block % 8) -- the
// rest the dX body: same arithmetic, same summation orders, bit-identical results.
template <typename H, int PK>
__global__ void __launch_bounds__(1024) k_bwd32_bf16(const int *__restrict__ table, int K, int identity_k, long long Rcap,
                                                     const long long *__restrict__ r_dev, const H *__restrict__ S,
                                                     const H *__restrict__ G, const float *__restrict__ W,
                                                     H *__restrict__ dX, float *__restrict__ part, int ngroups,
                                                     int nbx_dw, int n_dw_pad, int n_dx, int dx_threads) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int bid = (int)blockIdx.x;
    if (bid < n_dw_pad) {
        if (bid >= nbx_dw * ngroups) return;
        gdw32_bf16_body<H>(smem, bid % nbx_dw, bid / nbx_dw, nbx_dw, table, PK, K, identity_k, Rcap, r_dev, S, G, part,
                           ngroups);
    } else {
        if ((int)threadIdx.x >= dx_threads) return;          // the dX body was tuned for fewer waves per block
        gconv32_bf16_body<H, true, PK>(smem, bid - n_dw_pad, n_dx, dx_threads, table, 0, K, identity_k, Rcap, r_dev, G, W,
                                       nullptr, dX, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------ dW 32 x 2
// First layer (Cin = 2): the STATIONARY rows are the 32-channel ones (dY, read once, coalesced) and the 2-channel
// input rows are gathered through the by-output table:  part[chunk][k][c][b] = sum_i G[table[km(k)][i]][c] * S[i][b]
// with c in {0,1}; both forms below run on the matrix cores.

// bf16 form of the first-layer dW on the matrix cores.  Per 32-row tile the wave builds, in its own LDS region,
//   A [32 rows][64]: column k*2+c = channel c of the row gathered through table[km(k)][row] (2 x bf16 = one dword
//                    per (row, k); columns 54..63 are zero padding), row stride 33 dwords (conflict-free fill),
//   B [32 rows][32]: the stationary dY rows,
// and accumulates D[col][b] += sum_rows A[row][col] * B[row][b] with 4 x v_mfma_f32_32x32x16_bf16 (two 32-column
// halves x two 16-row steps; both operands are read column-wise from LDS, as in k_gdw32_bf16).  All of a tile's
// table reads are issued together, then all gathers: two memory latencies per tile, one tile per wave in flight.
constexpr int C2_WAVES = 8;

template <typename H>
__global__ void __launch_bounds__(512, 2) k_gdw_c32c2_bf16(const int *__restrict__ table, int mirror, int K,
                                                          int identity_k, long long Rcap,
                                                          const long long *__restrict__ r_dev,
                                                          const H *__restrict__ S, const H *__restrict__ G,
                                                          float *__restrict__ part) {
    __shared__ __attribute__((aligned(16))) unsigned sA[C2_WAVES][32 * 33];
    __shared__ __attribute__((aligned(16))) unsigned short sB[C2_WAVES][32 * 32];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int grow = lane >> 2, gchunk = lane & 3;
    unsigned *myA = sA[wid];
    unsigned short *myB = sB[wid];
    const long long R = valid_rows(Rcap, r_dev);
    const long long ntiles = (R + 31) >> 5;
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
    const unsigned *Gw = reinterpret_cast<const unsigned *>(G);          // one dword = the 2 bf16 channels of a row
    for (long long tile = (long long)blockIdx.x * C2_WAVES + wid; tile < ntiles; tile += (long long)gridDim.x * C2_WAVES) {
        const long long row0 = tile * 32;
        // ---- table entries: slot t covers offsets 2t (lanes 0..31) and 2t+1 (lanes 32..63), row = lane & 31
        const long long trow = row0 + c < R ? row0 + c : R - 1;
        const bool tlive = row0 + c < R;
        int nb[14];
#pragma unroll
        for (int t = 0; t < 14; ++t) {
            int k = 2 * t + h;
            int kk = k < K ? k : K - 1;
            nb[t] = table[(long long)(mirror ? K - 1 - kk : kk) * Rcap + trow];
        }
        // ---- stationary rows (issued before the gathers' addresses are known)
        const long long ra = row0 + grow, rb = row0 + grow + 16;
        uint4 s0 = *(const uint4 *)(S + (ra < R ? ra : R - 1) * 32 + gchunk * 8);
        uint4 s1 = *(const uint4 *)(S + (rb < R ? rb : R - 1) * 32 + gchunk * 8);
        unsigned xv[14];
#pragma unroll
        for (int t = 0; t < 14; ++t) {
            int k = 2 * t + h;
            int n = (k == identity_k) ? (int)trow : nb[t];
            bool ok = tlive && k < K && n >= 0;
            nb[t] = ok ? n : -1;
            xv[t] = Gw[ok ? n : 0];
        }
        __builtin_amdgcn_wave_barrier();            // the previous tile's fragment reads are done (LDS is in order)
#pragma unroll
        for (int t = 0; t < 14; ++t) {
            int k = 2 * t + h;
            myA[c * 33 + k] = nb[t] >= 0 ? xv[t] : 0u;          // k = 27 (h = 1, t = 13) lands in the zero padding
        }
        if (h == 0) {
#pragma unroll
            for (int k = 28; k < 32; ++k) myA[c * 33 + k] = 0u;
        }
        *(uint4 *)(myB + grow * 32 + gchunk * 8) = keep_if(s0, ra < R);
        *(uint4 *)(myB + (grow + 16) * 32 + gchunk * 8) = keep_if(s1, rb < R);
        __builtin_amdgcn_wave_barrier();
        // ---- fragments: A^T (columns of the [row][66 u16] image), B (columns of the [row][32] image)
        const unsigned short *A16 = reinterpret_cast<const unsigned short *>(myA);
        unsigned wa[2][2][4], wb[2][4];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                int r0 = 16 * st + 8 * h + 2 * m;
                wa[0][st][m] = (unsigned)A16[r0 * 66 + c] | ((unsigned)A16[(r0 + 1) * 66 + c] << 16);
                wa[1][st][m] = (unsigned)A16[r0 * 66 + 32 + c] | ((unsigned)A16[(r0 + 1) * 66 + 32 + c] << 16);
                wb[st][m] = (unsigned)myB[r0 * 32 + c] | ((unsigned)myB[(r0 + 1) * 32 + c] << 16);
            }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            uint4 a_lo = {wa[0][st][0], wa[0][st][1], wa[0][st][2], wa[0][st][3]};
            uint4 a_hi = {wa[1][st][0], wa[1][st][1], wa[1][st][2], wa[1][st][3]};
            uint4 bb = {wb[st][0], wb[st][1], wb[st][2], wb[st][3]};
            acc0 = mfma16<H>(a_lo, bb, acc0);
            acc1 = mfma16<H>(a_hi, bb, acc1);
        }
    }
    // deterministic block reduction in two halves (offsets 0..15, 16..31): every wave parks its accumulator in the
    // free A-staging area (8 x 4 KiB), then each thread adds output elements over the waves in wave order
    __syncthreads();
    float *sRed = reinterpret_cast<float *>(&sA[0][0]);                  // [C2_WAVES][1024] <= 8 x 32 x 33 words
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int col = (i & 3) + 8 * (i >> 2) + 4 * h;
            sRed[wid * 1024 + col * 32 + c] = half ? acc1[i] : acc0[i];
        }
        __syncthreads();
        // part[block][k][ch][b] = element (k*2 + ch, b) for the first 2K columns
        for (int e = threadIdx.x; e < 1024; e += 512) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < C2_WAVES; ++w) v += sRed[w * 1024 + e];
            const int ee = half * 1024 + e;
            if (ee < K * 64) part[(long long)blockIdx.x * K * 64 + ee] = v;
        }
        __syncthreads();
    }
}

// fp32 form of the first-layer dW on the matrix cores (exact fp32, v_mfma_f32_32x32x2_f32 with the tile's ROWS as the
// contraction): D[col = 2 k + ch][b] += sum_rows Xg[row][col] * dY[row][b].  Lane (c, h) supplies, for row pair s,
// A = Xg[2 s + h][col = c (+ 32)] and B = dY[2 s + h][b = c] -- the dY rows straight from global memory (whole 128-byte
// rows), the gathered 2-channel rows through the tile's table entries, which are read coalesced and turned around
// through a padded LDS image.  32 MFMAs per tile, one tile per wave, two memory round trips per tile.
constexpr int C2F_WAVES = 8;
__global__ void __launch_bounds__(512, 2) k_gdw_c32c2_f32(const int *__restrict__ table, int mirror, int K, int identity_k,
                                                         long long Rcap, const long long *__restrict__ r_dev,
                                                         const float *__restrict__ S, const float *__restrict__ G,
                                                         float *__restrict__ part) {
    __shared__ int sNbT[C2F_WAVES][28 * 33];                     // [k][row], row stride 33: conflict-free both ways
    __shared__ float sRed[C2F_WAVES][1024];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    int *myNb = sNbT[wid];
    const long long R = valid_rows(Rcap, r_dev);
    const long long ntiles = (R + 31) >> 5;
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
    // the two columns this lane feeds: col0 = c (offset c >> 1), col1 = 32 + c (offset 16 + (c >> 1)), channel c & 1
    const int k0 = c >> 1, k1 = 16 + (c >> 1), ch = c & 1;
    for (long long tile = (long long)blockIdx.x * C2F_WAVES + wid; tile < ntiles; tile += (long long)gridDim.x * C2F_WAVES) {
        const long long row0 = tile * 32;
        const long long trow = row0 + c < R ? row0 + c : R - 1;
        const bool tlive = row0 + c < R;
        int nb[14];
#pragma unroll
        for (int t = 0; t < 14; ++t) {
            const int k = 2 * t + h;
            const int kk = k < K ? k : K - 1;
            nb[t] = table[(long long)(mirror ? K - 1 - kk : kk) * Rcap + trow];
        }
        float bv[16];
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const long long row = row0 + 2 * s2 + h;
            const float t = S[(row < R ? row : R - 1) * 32 + c];
            bv[s2] = row < R ? t : 0.f;
        }
        __builtin_amdgcn_wave_barrier();            // the previous tile's reads of the image are done (LDS is in order)
#pragma unroll
        for (int t = 0; t < 14; ++t) {
            const int k = 2 * t + h;
            const int n = (k == identity_k) ? (int)trow : nb[t];
            myNb[k * 33 + c] = (tlive && k < K && n >= 0) ? n : -1;       // k = 27 (h = 1, t = 13): never read as valid
        }
        __builtin_amdgcn_wave_barrier();
        float x0[16], x1[16];
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const int ri = 2 * s2 + h;
            const int n0 = myNb[k0 * 33 + ri];
            const int n1 = k1 < 28 ? myNb[(k1 < 28 ? k1 : 27) * 33 + ri] : -1;
            const float v0 = G[(long long)(n0 >= 0 ? n0 : 0) * 2 + ch];
            const float v1 = G[(long long)(n1 >= 0 ? n1 : 0) * 2 + ch];
            x0[s2] = n0 >= 0 ? v0 : 0.f;
            x1[s2] = (n1 >= 0 && k1 < K) ? v1 : 0.f;
        }
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[s2], bv[s2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[s2], bv[s2], acc1, 0, 0, 0);
        }
    }
    // deterministic block reduction in two halves (columns 0..31, 32..63), waves added in wave order
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int col = (i & 3) + 8 * (i >> 2) + 4 * h;
            sRed[wid][col * 32 + c] = half ? acc1[i] : acc0[i];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 1024; e += 512) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < C2F_WAVES; ++w) v += sRed[w][e];
            const int ee = half * 1024 + e;                       // = (2 k + ch) * 32 + b
            if (ee < K * 64) part[(long long)blockIdx.x * K * 64 + ee] = v;
        }
        __syncthreads();
    }
}

// dW[k][a][b] (swap==0) or dW[k][b][a] (swap==1) = sum over slabs of part[slab][k][a][b]; blockDim.x / 32 slab slices
// per output (8, or 32 when there are many slabs) are summed in parallel and folded in slice order (deterministic).
__global__ void __launch_bounds__(1024) k_slab_reduce(const float *__restrict__ part, long long nslabs, long long per,
                                                      int K, int Cs, int Cg, int swap, float *__restrict__ dW) {
    __shared__ float sR[32][32];
    const int nsl = blockDim.x >> 5;
    const int sl = threadIdx.x >> 5, lane = threadIdx.x & 31;
    const long long e = (long long)blockIdx.x * 32 + lane;
    float s = 0.f;
    if (e < per)
        for (long long c = sl; c < nslabs; c += nsl) s += part[c * per + e];
    sR[sl][lane] = s;
    __syncthreads();
    if (sl == 0 && e < per) {
        s = 0.f;
        for (int q = 0; q < nsl; ++q) s += sR[q][lane];
        if (swap) {
            int k = (int)(e / ((long long)Cs * Cg));
            int rem = (int)(e % ((long long)Cs * Cg));
            int a = rem / Cg, b = rem % Cg;
            dW[((long long)k * Cg + b) * Cs + a] = s;
        } else {
            dW[e] = s;
        }
    }
}

// The same for several pending reductions in one launch (wfs_dw_reduce_jobs): block b serves job j with
// first[j] <= b < first[j + 1], 32 outputs per block; bitwise the results of k_slab_reduce (same slices, same order).
constexpr int MAX_DW_JOBS = 16;
struct DwJobs {
    wfs_dw_job j[MAX_DW_JOBS];
    int first[MAX_DW_JOBS + 1];
    int n;
};
__global__ void __launch_bounds__(1024) k_slab_reduce_multi(DwJobs jobs) {
    __shared__ float sR[32][32];
    int ji = 0;
    while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.first[ji + 1]) ++ji;
    const wfs_dw_job &jb = jobs.j[ji];
    const int nsl = jb.nslabs > 64 ? 32 : 8;          // the slicing (= summation order) of the single-job launches
    const int groups = 32 / nsl;                      // a block serves `groups` x 32 outputs with nsl slices each
    const int row = threadIdx.x >> 5, lane = threadIdx.x & 31;
    const int grp = row / nsl, sl = row % nsl;
    const long long e = ((long long)((int)blockIdx.x - jobs.first[ji]) * groups + grp) * 32 + lane;
    float s = 0.f;
    if (e < jb.per)
        for (long long c = sl; c < jb.nslabs; c += nsl) s += jb.part[c * jb.per + e];
    sR[row][lane] = s;
    __syncthreads();
    if (sl == 0 && e < jb.per) {
        s = 0.f;
        for (int q = 0; q < nsl; ++q) s += sR[grp * nsl + q][lane];
        if (jb.transpose) {
            const long long ab = (long long)jb.A * jb.B;
            const int k = (int)(e / ab), rem = (int)(e % ab);
            const int a = rem / jb.B, b = rem % jb.B;
            jb.dW[((long long)k * jb.B + b) * jb.A + a] = s;
        } else {
            jb.dW[e] = s;
        }
    }
}

// The same with FOUR neighbouring outputs per thread (16-byte loads, all of a thread's slabs asked for before the first
// add): same slices, same order per output -- bitwise the results above -- with a quarter of the threads.  Every job's
// `per` must be a multiple of 4 (the launcher checks).
__global__ void __launch_bounds__(1024) k_slab_reduce_multi4(DwJobs jobs) {
    __shared__ f32x4 sR4[32][32];
    int ji = 0;
    while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.first[ji + 1]) ++ji;
    const wfs_dw_job &jb = jobs.j[ji];
    const int nsl = jb.nslabs > 64 ? 32 : 8;
    const int groups = 32 / nsl;
    const int row = threadIdx.x >> 5, lane = threadIdx.x & 31;
    const int grp = row / nsl, sl = row % nsl;
    const long long e = (((long long)((int)blockIdx.x - jobs.first[ji]) * groups + grp) * 32 + lane) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (e < jb.per) {
        constexpr int U = 6;                          // slabs in flight per thread
        for (long long c0 = sl; c0 < jb.nslabs; c0 += (long long)nsl * U) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long c = c0 + (long long)u * nsl;
                const long long cc = c < jb.nslabs ? c : sl;          // clamped: the load is unconditional
                v[u] = *(const f32x4 *)(jb.part + cc * jb.per + e);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (c0 + (long long)u * nsl < jb.nslabs) s += v[u];
        }
    }
    sR4[row][lane] = s;
    __syncthreads();
    if (sl == 0 && e < jb.per) {
        s = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < nsl; ++q) s += sR4[grp * nsl + q][lane];
        if (jb.transpose) {
            const long long ab = (long long)jb.A * jb.B;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long long ei = e + i;
                const int k = (int)(ei / ab), rem = (int)(ei % ab);
                const int a = rem / jb.B, b = rem % jb.B;
                jb.dW[((long long)k * jb.B + b) * jb.A + a] = s[i];
            }
        } else if ((reinterpret_cast<uintptr_t>(jb.dW + e) & 15) == 0) {
            *(f32x4 *)(jb.dW + e) = s;
        } else {                                       // a slot of the flat gradient buffer at any 4-byte offset
#pragma unroll
            for (int i = 0; i < 4; ++i) jb.dW[e + i] = s[i];
        }
    }
}

}  // namespace

int wfs_launch_dw_jobs(const wfs_dw_job *jobs, int n, hipStream_t stream) {
    DwJobs dj;
    dj.n = n;
    bool by4 = true;
    for (int i = 0; i < n; ++i)
        by4 = by4 && jobs[i].per % 4 == 0 && ((uintptr_t)jobs[i].part & 15) == 0;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        dj.j[i] = jobs[i];
        dj.first[i] = blocks;
        const long long per_block = (jobs[i].nslabs > 64 ? 32 : 128) * (by4 ? 4 : 1);       // see k_slab_reduce_multi
        blocks += (int)((jobs[i].per + per_block - 1) / per_block);
    }
    dj.first[n] = blocks;
    if (blocks == 0) return WFS_OK;
    if (by4)
        k_slab_reduce_multi4<<<dim3((unsigned)blocks), dim3(1024), 0, stream>>>(dj);
    else
        k_slab_reduce_multi<<<dim3((unsigned)blocks), dim3(1024), 0, stream>>>(dj);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

// fp32 rows of the 32 -> 32 layers as three bf16 pieces on the bf16 matrix cores (k_gdw32_split, k_gconv16_split);
// WFS_SPLIT_BF16=0 keeps the fp32 matrix instructions (read once)
static bool wfs_split_bf16() {
    static const int on = [] {
        const char *e = getenv("WFS_SPLIT_BF16");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    return on != 0;
}

// ---- launchers used by gather_conv.hip's C entry points -------------------------------------------------
bool wfs_mfma_gconv32_ok(int K) { return K >= 1 && K <= 32; }       // K * 4 KiB of LDS <= 128 KiB

template <typename KernelT, typename... Args>
static int launch_big_lds(KernelT kernel, bool *attr_done, dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                          Args... args) {
    if (!*attr_done) {
        // 160 KiB per CU minus the kernels' static LDS
        WFS_HIP_CHECK(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
        *attr_done = true;
    }
    kernel<<<grid, block, lds, stream>>>(args...);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

// grid of the two 32 -> 32 kernels: <= 256 persistent blocks (multiple of 8: one slice of the row range per XCD)
static void gconv32_grid(long long R, bool padded, long long *ntiles, int *wpb, long long *nblk, long long *tiles_per_xcd,
                         int max_waves) {
    *ntiles = (R + 31) >> 5;
    // waves per block: enough tiles per SIMD without leaving CUs idle on small inputs.  With a device-side row count R is
    // a capacity (a captured step pads by 10 - 40 %) and the kernels share out the VALID tiles: size the block for 7/8 of
    // the capacity -- at the PSD batch 12 waves instead of 16 (0.5373 -> 0.526 ms per step at the default headroom);
    // should more tiles be valid than wave slots exist, some waves take a second tile
    const long long expect = padded ? (*ntiles * 7 + 7) / 8 : *ntiles;
    int w = (int)((expect + 255) / 256);
    w = w < 4 ? 4 : (w > max_waves ? max_waves : (w + 3) / 4 * 4);
    long long nb = (expect + w - 1) / w;
    if (nb > 256) nb = 256;
    nb = (nb + 7) / 8 * 8;
    *wpb = w;
    *nblk = nb;
    *tiles_per_xcd = (*ntiles + 7) / 8;
}

int wfs_launch_gconv32_f32(const int *table, int mirror, int K, int identity_k, long long R, const long long *r_dev,
                           const float *X, const float *W, int transpose_w, const float *bias, float *Y,
                           hipStream_t stream, int packed_kl) {
    const size_t lds = (size_t)K * 4096;
    WFS_REQUIRE(packed_kl == 0 || (packed_kl == 3 && transpose_w && !mirror && identity_k < 0), WFS_EINVAL,
                "packed tables: kl = 3, dX products only");
    // 16-row tiles taken off a per-block counter (k_gconv16_f32): up to 16 waves per block, <= 256 blocks
    static bool attr16[2] = {false, false};
    const long long nt16 = (R + 15) >> 4;
    const long long expect = r_dev ? (nt16 * 7 + 7) / 8 : nt16;
    int w = (int)((expect + 255) / 256);
    w = w < 4 ? 4 : (w > 16 ? 16 : (w + 3) / 4 * 4);
    long long nb = (expect + w - 1) / w;
    nb = nb > 256 ? 256 : (nb + 7) / 8 * 8;
    if (nb < 8) nb = 8;
    if (wfs_split_bf16() && K <= 27) {
        // 12 waves at most (the pieces cost registers), K - 1 offsets of 6 KiB in LDS
        int ws = w > 12 ? 12 : w;
        long long nbs = (expect + ws - 1) / ws;
        nbs = nbs > 256 ? 256 : (nbs + 7) / 8 * 8;
        if (nbs < 8) nbs = 8;
        const dim3 gs((unsigned)nbs), bs(ws * 64);
        const size_t lds_s = (size_t)(K - 1) * 6144;
        static bool attrs[3] = {false, false, false};
        if (packed_kl)
            return launch_big_lds(k_gconv16_split<true, 3>, &attrs[2], gs, bs, lds_s, stream, table, mirror, K, identity_k, R,
                                  r_dev, X, W, bias, Y);
        if (transpose_w)
            return launch_big_lds(k_gconv16_split<true>, &attrs[0], gs, bs, lds_s, stream, table, mirror, K, identity_k, R,
                                  r_dev, X, W, bias, Y);
        return launch_big_lds(k_gconv16_split<false>, &attrs[1], gs, bs, lds_s, stream, table, mirror, K, identity_k, R, r_dev,
                              X, W, bias, Y);
    }
    const dim3 g16((unsigned)nb), b16(w * 64);
    if (packed_kl) {
        static bool attr16p = false;
        return launch_big_lds(k_gconv16_f32<true, 3>, &attr16p, g16, b16, lds, stream, table, mirror, K, identity_k, R,
                              r_dev, X, W, bias, Y);
    }
    if (transpose_w)
        return launch_big_lds(k_gconv16_f32<true>, &attr16[0], g16, b16, lds, stream, table, mirror, K, identity_k, R,
                              r_dev, X, W, bias, Y);
    return launch_big_lds(k_gconv16_f32<false>, &attr16[1], g16, b16, lds, stream, table, mirror, K, identity_k, R,
                          r_dev, X, W, bias, Y);
}

template <typename H>
static int launch_gconv32_h16(const int *table, int mirror, int K, int identity_k, long long R, const long long *r_dev,
                              const H *Xb, const float *W, int transpose_w, const float *bias, H *Yb,
                              hipStream_t stream, int packed_kl) {
    long long ntiles, nblk, tiles_per_xcd;
    int wpb;
    gconv32_grid(R, r_dev != nullptr, &ntiles, &wpb, &nblk, &tiles_per_xcd, 16);
    const size_t lds = (size_t)K * 2048 + (size_t)wpb * K * 32 * sizeof(int);
    static bool attr[2] = {false, false};          // per instantiation of this template, i.e. per H
    const dim3 grid((unsigned)nblk), block(wpb * 64);
    WFS_REQUIRE(packed_kl == 0 || (packed_kl == 3 && transpose_w && !mirror && identity_k < 0), WFS_EINVAL,
                "packed tables: kl = 3, dX products only");
    if (packed_kl) {
        static bool attr_p = false;
        return launch_big_lds(k_gconv32_bf16<H, true, 3>, &attr_p, grid, block, lds, stream, table, mirror, K, identity_k, R,
                              r_dev, Xb, W, bias, Yb, ntiles, tiles_per_xcd);
    }
    if (transpose_w)
        return launch_big_lds(k_gconv32_bf16<H, true>, &attr[0], grid, block, lds, stream, table, mirror, K, identity_k, R,
                              r_dev, Xb, W, bias, Yb, ntiles, tiles_per_xcd);
    return launch_big_lds(k_gconv32_bf16<H, false>, &attr[1], grid, block, lds, stream, table, mirror, K, identity_k, R,
                          r_dev, Xb, W, bias, Yb, ntiles, tiles_per_xcd);
}

int wfs_launch_gconv32_h16(const int *table, int mirror, int K, int identity_k, long long R, const long long *r_dev,
                           const void *X, const float *W, int transpose_w, const float *bias, void *Y, int dtype,
                           hipStream_t stream, int packed_kl) {
    if (dtype == WFS_F16)
        return launch_gconv32_h16<wfs_f16>(table, mirror, K, identity_k, R, r_dev, (const wfs_f16 *)X, W, transpose_w,
                                           bias, (wfs_f16 *)Y, stream, packed_kl);
    return launch_gconv32_h16<wfs_bf16>(table, mirror, K, identity_k, R, r_dev, (const wfs_bf16 *)X, W, transpose_w, bias,
                                        (wfs_bf16 *)Y, stream, packed_kl);
}

// 2 -> 32
int wfs_launch_gconv_c2c32(const int *table, const int *kmap, int K, int identity_k, long long R,
                           const long long *r_dev, const void *X, const float *W, const float *bias, void *Y, int dtype,
                           hipStream_t stream) {
    KMap km;
    bool is_ident = true, is_mirror = true;
    for (int k = 0; k < K; ++k) {
        km.v[k] = kmap ? kmap[k] : k;
        is_ident = is_ident && km.v[k] == k;
        is_mirror = is_mirror && km.v[k] == K - 1 - k;
    }
    if (dtype != WFS_F32 && K <= 32 && (is_ident || is_mirror)) {
        long long nb = ((R + 31) / 32 + 3) / 4;              // one 32-row tile per wave, 4 waves per block
        if (nb > 4096) nb = 4096;
        const dim3 grid((unsigned)nb), block(256);
        const int mir = is_ident ? 0 : 1;
        if (dtype == WFS_F16)
            k_gconv_c2c32_bf16<wfs_f16><<<grid, block, 0, stream>>>(table, mir, K, identity_k, R, r_dev, (const wfs_f16 *)X, W,
                                                                    bias, (wfs_f16 *)Y);
        else
            k_gconv_c2c32_bf16<wfs_bf16><<<grid, block, 0, stream>>>(table, mir, K, identity_k, R, r_dev, (const wfs_bf16 *)X,
                                                                     W, bias, (wfs_bf16 *)Y);
        WFS_LAUNCH_CHECK();
        return WFS_OK;
    }
    if (dtype == WFS_F32 && K <= 27 && (is_ident || is_mirror)) {
        long long nb = ((R + 31) / 32 + 3) / 4;              // one 32-row tile per wave, 4 waves per block
        if (nb > 4096) nb = 4096;
        k_gconv_c2c32_f32<<<dim3((unsigned)nb), dim3(256), 0, stream>>>(table, is_ident ? 0 : 1, K, identity_k, R, r_dev,
                                                                        (const float *)X, W, bias, (float *)Y);
        WFS_LAUNCH_CHECK();
        return WFS_OK;
    }
    long long nblk = (R + 31) / 32;
    if (nblk > 8192) nblk = 8192;
    if (dtype == WFS_F32)
        k_gconv_c2c32<float><<<dim3((unsigned)nblk), dim3(256), 0, stream>>>(table, km, K, identity_k, R, r_dev,
                                                                            (const float *)X, W, bias, (float *)Y);
    else if (dtype == WFS_BF16)
        k_gconv_c2c32<wfs_bf16><<<dim3((unsigned)nblk), dim3(256), 0, stream>>>(
            table, km, K, identity_k, R, r_dev, (const wfs_bf16 *)X, W, bias, (wfs_bf16 *)Y);
    else
        k_gconv_c2c32<wfs_f16><<<dim3((unsigned)nblk), dim3(256), 0, stream>>>(
            table, km, K, identity_k, R, r_dev, (const wfs_f16 *)X, W, bias, (wfs_f16 *)Y);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

// slabs for the 32x32 dW: blocks over rows; returns the slab count through *nslabs
static long long dw32_blocks(long long R, bool two_per_cu = false) {
    long long ntiles = (R + 31) >> 5;
    long long nblk = (ntiles + (two_per_cu ? 19 : 39)) / (two_per_cu ? 20 : 40);        // ~40 (20) tiles per block
    if (nblk < 1) nblk = 1;
    const long long cap = two_per_cu ? 72 : 36; // x 7 offset sets = 252 (504) blocks: one round on 256 CUs, one (two) per CU
    if (nblk > cap) nblk = cap;
    return nblk;
}
static long long dwc2_chunks(long long R) {
    long long c = (R + 127) / 128;
    if (c < 1) c = 1;
    if (c > 512) c = 512;
    return c;
}

size_t wfs_dw_fast_workspace(int K, long long R, int Cs, int Cg) {
    if (Cs == 32 && Cg == 32) return (size_t)dw32_blocks(R, true) * K * 1024 * sizeof(float);   // the larger (fp32) grid
    if (Cs == 32 && Cg == 2) return (size_t)dwc2_chunks(R) * K * 64 * sizeof(float);
    return 0;
}

int wfs_launch_gdw32(const int *table, int K, int identity_k, long long R, const long long *r_dev, const void *S,
                     const void *G, int swap, float *dW, float *part, int dtype, wfs_dw_job *defer, hipStream_t stream,
                     int packed_kl) {
    WFS_REQUIRE(packed_kl == 0 || (packed_kl >= 1 && packed_kl <= 8 && K % packed_kl == 0 && identity_k < 0), WFS_EINVAL,
                "packed tables: K must be a multiple of kl <= 8, no identity offset");
    const long long nblk = dw32_blocks(R, dtype == WFS_F32 && !wfs_split_bf16());
    const long long ntiles = (R + 31) >> 5;
    const long long tiles_per_block = (ntiles + nblk - 1) / nblk;
    const int ngroups = (K + DW_KG - 1) / DW_KG;
    const dim3 grid((unsigned)nblk, (unsigned)ngroups);
    if (dtype == WFS_F32 && wfs_split_bf16()) {
        static bool attr = false;
        const int rc = launch_big_lds(k_gdw32_split, &attr, grid, dim3(DWS_WAVES * 64), DWS_LDS, stream, table, packed_kl, K, identity_k, R,
                                      r_dev, (const float *)S, (const float *)G, part, ngroups);
        if (rc != WFS_OK) return rc;
    } else if (dtype == WFS_F32) {
        k_gdw32<float><<<grid, dim3(512), 0, stream>>>(table, packed_kl, K, identity_k, R, r_dev, (const float *)S,
                                                       (const float *)G, part, ngroups, tiles_per_block);
        WFS_LAUNCH_CHECK();
    } else if (dtype == WFS_BF16) {
        static bool attr = false;
        const int rc = launch_big_lds(k_gdw32_bf16<wfs_bf16>, &attr, grid, dim3(1024), DWB_LDS, stream, table, packed_kl, K,
                                      identity_k, R, r_dev, (const wfs_bf16 *)S, (const wfs_bf16 *)G, part, ngroups,
                                      tiles_per_block);
        if (rc != WFS_OK) return rc;
    } else {
        static bool attr = false;
        const int rc = launch_big_lds(k_gdw32_bf16<wfs_f16>, &attr, grid, dim3(1024), DWB_LDS, stream, table, packed_kl, K,
                                      identity_k, R, r_dev, (const wfs_f16 *)S, (const wfs_f16 *)G, part, ngroups,
                                      tiles_per_block);
        if (rc != WFS_OK) return rc;
    }
    const long long per = (long long)K * 1024;
    if (defer) {
        *defer = wfs_dw_job{part, nblk, per, K, 32, 32, swap, dW};
        return WFS_OK;
    }
    k_slab_reduce<<<dim3((unsigned)((per + 31) / 32)), dim3(nblk > 64 ? 1024 : 256), 0, stream>>>(part, nblk, per, K, 32, 32, swap, dW);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

// dW and dX of a 32 -> 32 layer with 16-bit rows in one launch (k_bwd32_bf16)
template <typename H>
static int launch_bwd32_h16(const int *table, int packed_kl, int K, int identity_k, long long R, const long long *r_dev,
                            const H *S, const H *G, const float *W, H *dX, float *part, hipStream_t stream) {
    const long long nbx_dw = dw32_blocks(R, false);
    const int ngroups = (K + DWB_KG - 1) / DWB_KG;
    const long long n_dw = nbx_dw * ngroups, n_dw_pad = (n_dw + 7) / 8 * 8;
    long long ntiles, n_dx, tiles_per_xcd;
    int wpb;
    gconv32_grid(R, r_dev != nullptr, &ntiles, &wpb, &n_dx, &tiles_per_xcd, 16);
    size_t lds = (size_t)K * 2048 + (size_t)wpb * K * 32 * sizeof(int);
    if (lds < (size_t)DWB_LDS) lds = DWB_LDS;
    const dim3 grid((unsigned)(n_dw_pad + n_dx)), block(1024);
    if (packed_kl) {
        static bool attr = false;
        return launch_big_lds(k_bwd32_bf16<H, 3>, &attr, grid, block, lds, stream, table, K, identity_k, R, r_dev, S, G, W, dX,
                              part, ngroups, (int)nbx_dw, (int)n_dw_pad, (int)n_dx, wpb * 64);
    }
    static bool attr0 = false;
    return launch_big_lds(k_bwd32_bf16<H, 0>, &attr0, grid, block, lds, stream, table, K, identity_k, R, r_dev, S, G, W, dX,
                          part, ngroups, (int)nbx_dw, (int)n_dw_pad, (int)n_dx, wpb * 64);
}

// 16-bit rows and three-piece fp32 rows.  The fp32-INSTRUCTION kernels were measured too (same construction: 0.828 vs
// 0.808 ms per captured step) -- that dW body is tuned for two 512-thread blocks per CU with 32 KB of LDS each, which it
// loses inside a 1024-thread, 110-KB launch
bool wfs_bwd32_fused_ok(int K, int packed_kl, int dtype) {
    const bool row_type = dtype == WFS_BF16 || dtype == WFS_F16 || (dtype == WFS_F32 && wfs_split_bf16());
    return row_type && K >= 1 && K <= 27 && (packed_kl == 0 || (packed_kl == 3 && K % 3 == 0));
}

// grid of the three-piece conv: <= 12 waves per block, <= 256 blocks (multiple of 8)
static void gconv16_split_grid(long long R, bool padded, int *waves, long long *nblk) {
    const long long nt16 = (R + 15) >> 4;
    const long long expect = padded ? (nt16 * 7 + 7) / 8 : nt16;
    int w = (int)((expect + 255) / 256);
    w = w < 4 ? 4 : (w > 12 ? 12 : (w + 3) / 4 * 4);
    long long nb = (expect + w - 1) / w;
    nb = nb > 256 ? 256 : (nb + 7) / 8 * 8;
    if (nb < 8) nb = 8;
    *waves = w;
    *nblk = nb;
}

static int launch_bwd32_split(const int *table, int packed_kl, int K, int identity_k, long long R, const long long *r_dev,
                              const float *S, const float *G, const float *W, float *dX, float *part, hipStream_t stream) {
    const long long nbx_dw = dw32_blocks(R, false);
    const int ngroups = (K + DWS_KG - 1) / DWS_KG;
    const long long n_dw = nbx_dw * ngroups, n_dw_pad = (n_dw + 7) / 8 * 8;
    int ws;
    long long n_dx;
    gconv16_split_grid(R, r_dev != nullptr, &ws, &n_dx);
    size_t lds = (size_t)(K - 1) * 6144;
    if (lds < (size_t)DWS_LDS) lds = DWS_LDS;
    const dim3 grid((unsigned)(n_dw_pad + n_dx)), block(DWS_WAVES * 64);
    if (packed_kl) {
        static bool attr = false;
        return launch_big_lds(k_bwd32_split<3>, &attr, grid, block, lds, stream, table, K, identity_k, R, r_dev, S, G, W, dX,
                              part, ngroups, (int)nbx_dw, (int)n_dw_pad, (int)n_dx, ws * 64);
    }
    static bool attr0 = false;
    return launch_big_lds(k_bwd32_split<0>, &attr0, grid, block, lds, stream, table, K, identity_k, R, r_dev, S, G, W, dX,
                          part, ngroups, (int)nbx_dw, (int)n_dw_pad, (int)n_dx, ws * 64);
}

int wfs_launch_bwd32_h16(const int *table, int packed_kl, int K, int identity_k, long long R, const long long *r_dev,
                         const void *S, const void *G, const float *W, void *dX, int swap, float *dW, float *part, int dtype,
                         wfs_dw_job *defer, hipStream_t stream) {
    int rc;
    if (dtype == WFS_F32)
        rc = launch_bwd32_split(table, packed_kl, K, identity_k, R, r_dev, (const float *)S, (const float *)G, W, (float *)dX,
                                part, stream);
    else if (dtype == WFS_F16)
        rc = launch_bwd32_h16<wfs_f16>(table, packed_kl, K, identity_k, R, r_dev, (const wfs_f16 *)S, (const wfs_f16 *)G, W,
                                       (wfs_f16 *)dX, part, stream);
    else
        rc = launch_bwd32_h16<wfs_bf16>(table, packed_kl, K, identity_k, R, r_dev, (const wfs_bf16 *)S, (const wfs_bf16 *)G,
                                        W, (wfs_bf16 *)dX, part, stream);
    if (rc != WFS_OK) return rc;
    const long long nblk = dw32_blocks(R, false);
    const long long per = (long long)K * 1024;
    if (defer) {
        *defer = wfs_dw_job{part, nblk, per, K, 32, 32, swap, dW};
        return WFS_OK;
    }
    k_slab_reduce<<<dim3((unsigned)((per + 31) / 32)), dim3(nblk > 64 ? 1024 : 256), 0, stream>>>(part, nblk, per, K, 32, 32, swap, dW);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

int wfs_launch_gdw_c32c2(const int *table, int mirror, int K, int identity_k, long long R, const long long *r_dev,
                         const void *S, const void *G, int swap, float *dW, float *part, int dtype,
                         wfs_dw_job *defer, hipStream_t stream) {
    long long chunks = dwc2_chunks(R);
    if (dtype == WFS_F32) {
        const long long ntiles = (R + 31) >> 5;
        chunks = (ntiles + C2F_WAVES - 1) / C2F_WAVES;        // one tile per wave, at most 512 blocks (= slabs)
        if (chunks > 512) chunks = 512;
        k_gdw_c32c2_f32<<<dim3((unsigned)chunks), dim3(512), 0, stream>>>(table, mirror, K, identity_k, R, r_dev,
                                                                          (const float *)S, (const float *)G, part);
    } else {
        const long long ntiles = (R + 31) >> 5;
        chunks = (ntiles + C2_WAVES - 1) / C2_WAVES;          // one tile per wave, at most 512 blocks (= slabs)
        if (chunks > 512) chunks = 512;
        if (dtype == WFS_BF16)
            k_gdw_c32c2_bf16<wfs_bf16><<<dim3((unsigned)chunks), dim3(512), 0, stream>>>(
                table, mirror, K, identity_k, R, r_dev, (const wfs_bf16 *)S, (const wfs_bf16 *)G, part);
        else
            k_gdw_c32c2_bf16<wfs_f16><<<dim3((unsigned)chunks), dim3(512), 0, stream>>>(
                table, mirror, K, identity_k, R, r_dev, (const wfs_f16 *)S, (const wfs_f16 *)G, part);
    }
    WFS_LAUNCH_CHECK();
    // part is [chunk][k][c (gathered, 2)][b (stationary, 32)] = the "swap" orientation of (S=32, G=2)
    const long long per = (long long)K * 64;
    if (defer) {
        *defer = wfs_dw_job{part, chunks, per, K, 2, 32, swap ? 0 : 1, dW};
        return WFS_OK;
    }
    k_slab_reduce<<<dim3((unsigned)((per + 31) / 32)), dim3(chunks > 64 ? 1024 : 256), 0, stream>>>(part, chunks, per, K, 2, 32, swap ? 0 : 1, dW);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}
