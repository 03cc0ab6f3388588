// evconv.hip -- EVENT-LOCAL sparse convolution kernels (round 3 experiment; NOT part of libwfsparse: verified bit-equal to the
// tile-parallel kernel, measured no faster inside the step -- see README.md here and profiles/r03_event_local_*).
//
// A sparse convolution never crosses events: the rulebook key includes the batch index, and the
// reference's collate_fn (src/engineering/PSDDataModule.py:10-20) concatenates the items of a batch in order, so the
// rows of one event are ONE contiguous range of every row set of the net -- the input voxels, and (first-seen numbering
// walks the inputs in order) the outputs of every regular conv.  These kernels use that:
//
//   (k_event_offsets, the row range of every event of an index set, lives in the product: csrc/evrulebook.hip)
//   k_slot_table      a gather table int32 [K, R] re-encoded per event: slot k of row r = 1 + (source row - first input
//                     row of r's event) as uint16, 0 = no neighbour; one 64-byte record per output row (32 slots)
//   k_evconv32        32 -> 32 channel gather conv (forward, dX with the transposed filter; SubM, regular, inverse --
//                     any table) on 16-bit rows: ONE WORKGROUP PER EVENT.  The event's input rows are streamed into
//                     LDS once (coalesced 16-B loads, ~21 KB for a PSD event) and every (offset, row) operand of the
//                     matrix cores is a conflict-free ds_read_b128 from there, instead of ~10 gathers of each row from
//                     L2 by the tile-parallel kernel (conv_mfma.hip k_gconv32_bf16: 41 MB of 64-byte L2 requests per
//                     launch at the PSD batch).  The launch is a chain of TWO dependent memory round trips: event
//                     descriptor -> {filters, rows, slot records}, all in flight together.
//
// Same arithmetic as k_gconv32_bf16: v_mfma_f32_32x32x16_bf16 / _f16, fp32 accumulate, filters rounded to the row type
// while they are staged; the order of the fp32 sum over offsets is the same (increasing k).
#include "wfs_common.h"
#include "event_local.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ long long valid_rows(long long R, const long long *r_dev) {
    long long v = r_dev ? *r_dev : R;
    return v < R ? v : R;
}

template <typename H, typename V>
__device__ __forceinline__ f32x16 mfma16(V a, V b, f32x16 acc) {
    static_assert(sizeof(V) == 16, "8 x 16-bit operands");
    if constexpr (__is_same(H, wfs_f16))
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0,
                                                      0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc,
                                                       0, 0, 0);
}

__device__ __forceinline__ uint4 keep_if(uint4 v, bool ok) {
    v.x = ok ? v.x : 0u;
    v.y = ok ? v.y : 0u;
    v.z = ok ? v.z : 0u;
    v.w = ok ? v.w : 0u;
    return v;
}

// ------------------------------------------------------------------------------------------ slot tables
// One 64-byte record per output row: 32 uint16 slots, slot k = 1 + (table[kmap k][row] - first input row of the row's
// event), 0 = no neighbour (and k >= K).  identity_k: the row itself (SubM centre; in / out offsets are the same array).
// Slots that do not fit 16 bits belong to events far beyond the LDS capacity of the consumers, which never read them.
__global__ void __launch_bounds__(256) k_slot_table(const int *__restrict__ table, int mirror, int K, int identity_k,
                                                    long long R, const long long *__restrict__ r_dev,
                                                    const int *__restrict__ out_ev, const int *__restrict__ in_ev, int B,
                                                    uint4 *__restrict__ ctab) {
    const int Rv = (int)valid_rows(R, r_dev);
    for (int e = blockIdx.x; e < B; e += gridDim.x) {
        const int o0 = out_ev[e], i0 = in_ev[e];
        int o1 = out_ev[e + 1];
        o1 = o1 < Rv ? o1 : Rv;
        for (int row = o0 + threadIdx.x; row < o1; row += 256) {
            unsigned s[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int kk = k < K ? k : K - 1;
                const int nb = (k == identity_k) ? row : table[(long long)(mirror ? K - 1 - kk : kk) * R + row];
                s[k] = (k < K && nb >= 0) ? (unsigned)(nb - i0 + 1) & 0xFFFFu : 0u;
            }
            uint4 *dst = ctab + (long long)row * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dst[q] = uint4{s[8 * q] | (s[8 * q + 1] << 16), s[8 * q + 2] | (s[8 * q + 3] << 16),
                               s[8 * q + 4] | (s[8 * q + 5] << 16), s[8 * q + 6] | (s[8 * q + 7] << 16)};
        }
    }
}

// ------------------------------------------------------------------------------------------ 32 -> 32, one event per block
// LDS:  sW  [K][s][h][col] fragments of 16 B (8 x 16-bit filter values; the image of conv_mfma.hip's k_gconv32_bf16)
//       sX  [cap + 1] rows at a stride of 80 B (64 B of data + 16 B of padding): the 16 lanes one ds_read_b128 serves per
//           cycle read 16 (mostly consecutive) rows at the same 16-byte chunk, and 5 * row mod 16 sends 16 consecutive
//           rows to 16 different bank quads.  Row 0 is all zeros -- slot 0, "no neighbour"; input row i0 + j is row j + 1.
// A tile = 32 consecutive output rows of ONE event (the last tile of an event is partial); a wave takes two at a time.
// timing knock-outs (make evknock; results are wrong by construction): 1 no MFMA / LDS operand reads, 2 no slot loads,
// 4 no row staging, 8 no filter staging, 16 no stores
#ifndef EV_KNOCK
#define EV_KNOCK 0
#endif
constexpr int EV_THREADS = 512;        // 8 waves, 2 per SIMD: 256 registers each
constexpr int EV_WAVES = EV_THREADS / 64;
constexpr int EV_ROWLOADS = 8;         // 16-byte row chunks per thread in flight at once: 1024 rows (larger events: more trips)
constexpr int EV_STRIDE = 80;

// MIRROR: the slot records are in the TABLE's offset order and the table is read mirrored (SubM forward through nbr_out:
// offset k gathers through entry 26 - k; K = 27 only, so that the entry index is a compile-time constant)
template <typename H, bool TRANSPOSE_W, bool MIRROR>
__global__ void __launch_bounds__(EV_THREADS) k_evconv32(const int *__restrict__ table, int mirror, int K, int identity_k,
                                                         long long R, const long long *__restrict__ r_dev,
                                                         const uint4 *__restrict__ ctab, const int *__restrict__ out_ev,
                                                         const int *__restrict__ in_ev, int B, const H *__restrict__ X,
                                                         const float *__restrict__ W, const float *__restrict__ bias,
                                                         H *__restrict__ Y, int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *sWb = reinterpret_cast<uint4 *>(smem);
    unsigned char *sXb = smem + (size_t)K * 2048;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;

    // is every row set grouped by event?  (each wave asks for itself: no barrier in front of the loads below)
    const int fl = out_ev[B + 1 + lane] | in_ev[B + 1 + lane];
    int e = blockIdx.x;
    int o0 = 0, o1 = 0, i0 = 0, i1 = 0;
    if (e < B) {
        o0 = out_ev[e];
        o1 = out_ev[e + 1];
        i0 = in_ev[e];
        i1 = in_ev[e + 1];
    }
    const int Rv = (int)valid_rows(R, r_dev);
    const bool structured = __ballot(fl != 0) == 0ull;

    // ---- filters: all of a thread's fragments are asked for before the first conversion
    constexpr int WSB = 7;          // 27 * 128 fragments over 512 threads
    const int nfrag = K * 128;
    float w[WSB][8];
#pragma unroll
    for (int b = 0; b < WSB; ++b) {
        int u = threadIdx.x + b * EV_THREADS;
        u = u < nfrag ? u : nfrag - 1;
        const int k = u >> 7, s = (u >> 6) & 1, hh = (u >> 5) & 1, col = u & 31;
        const int c0 = 16 * hh + 8 * s;
        if (!TRANSPOSE_W) {
#pragma unroll
            for (int j = 0; j < 8; ++j) w[b][j] = W[(k * 32 + c0 + j) * 32 + col];
        } else {
            const f32x4 *src = (const f32x4 *)(W + (k * 32 + col) * 32 + c0);
            const f32x4 lo = src[0], hi = src[1];
            w[b][0] = lo.x; w[b][1] = lo.y; w[b][2] = lo.z; w[b][3] = lo.w;
            w[b][4] = hi.x; w[b][5] = hi.y; w[b][6] = hi.z; w[b][7] = hi.w;
        }
    }
    const float bj = bias ? bias[r] : 0.f;

    // epilogue of one 32-row tile: reg i holds (row (i&3) + 8(i>>2) + 4h, col r); neighbouring columns are paired with
    // one lane exchange so that every lane stores one packed dword: even lanes row(i), odd lanes row(i+1)
    auto store_tile = [&](const f32x16 &acc, int g0, int row_end) {
        unsigned *Yw = reinterpret_cast<unsigned *>(Y) + (long long)g0 * 16 + (r >> 1);
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const float mine0 = acc[i], mine1 = acc[i + 1];
            const float send = (lane & 1) ? mine0 : mine1;
            const float got = __shfl_xor(send, 1, 64);
            const unsigned packed = (lane & 1) ? wfs_pack2<H>(got, mine1) : wfs_pack2<H>(mine0, got);
            const int ri = (lane & 1) ? i + 1 : i;
            const int orow = (ri & 3) + 8 * (ri >> 2) + 4 * h;
            if (!(EV_KNOCK & 16) || packed == 0x12345678u)
                if (g0 + orow < row_end) Yw[orow * 16] = packed;
        }
    };
    // LDS path.  A wave works on TWO neighbouring tiles at a time (rows g0 .. g0 + 63 of the output set): the filter
    // fragments of an offset are read from LDS once for both, and the two accumulator chains interleave on the matrix
    // pipe.  ca / cb: the slot records of the lane's two rows (g0 + r, g0 + 32 + r), 16 dwords of 2 slots each.
    uint4 ca[4], cb[4];
    auto load_slots = [&](int g0, int row_end) {     // callers give row_end > g0 >= 0
        const int row = g0 + r;
        const uint4 *pa = ctab + (long long)(row < row_end ? row : row_end - 1) * 4;
        const uint4 *pb = ctab + (long long)(row + 32 < row_end ? row + 32 : row_end - 1) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ca[q] = (EV_KNOCK & 2) ? uint4{1u, 1u, 1u, 1u} : pa[q];
            cb[q] = (EV_KNOCK & 2) ? uint4{1u, 1u, 1u, 1u} : pb[q];
        }
    };
    auto slot_of = [](const uint4 (&c)[4], int k) -> unsigned {
        const uint4 v = c[k >> 3];
        const int d = (k >> 1) & 3;
        const unsigned wd = d == 0 ? v.x : (d == 1 ? v.y : (d == 2 ? v.z : v.w));
        return (k & 1) ? (wd >> 16) : (wd & 0xFFFFu);
    };
    auto tile_lds = [&](int g0, int row_end) {
        const bool livea = g0 + r < row_end, liveb = g0 + 32 + r < row_end;
        if (!livea) ca[0] = ca[1] = ca[2] = ca[3] = uint4{0u, 0u, 0u, 0u};
        if (!liveb) cb[0] = cb[1] = cb[2] = cb[3] = uint4{0u, 0u, 0u, 0u};
        f32x16 acca, accb;
#pragma unroll
        for (int i = 0; i < 16; ++i) acca[i] = accb[i] = bj;
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            // does any row of the two tiles use one of the group's three offsets?
            unsigned any = 0;
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) any |= slot_of(ca, MIRROR ? 26 - (3 * g + kk) : 3 * g + kk) | slot_of(cb, MIRROR ? 26 - (3 * g + kk) : 3 * g + kk);
            if ((EV_KNOCK & 1) || __ballot(any != 0u) == 0ull) continue;
            // all operands of the group's three offsets are asked for before the first MFMA
            uint4 b0[3], b1[3], alo[3], ahi[3], blo[3], bhi[3];
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                const int k = 3 * g + kk;
                const int kc = k < K ? k : K - 1;          // k >= K: slots are 0 (zero rows), any filter will do
                const uint4 *bp = sWb + kc * 128 + h * 32 + r;
                b0[kk] = bp[0];
                b1[kk] = bp[64];
                const int ke = MIRROR ? 26 - k : k;
                const unsigned char *xa = sXb + ((EV_KNOCK & 256) ? 0u : slot_of(ca, ke)) * EV_STRIDE + 32 * h;
                const unsigned char *xb = sXb + ((EV_KNOCK & 256) ? 0u : slot_of(cb, ke)) * EV_STRIDE + 32 * h;
                alo[kk] = *reinterpret_cast<const uint4 *>(xa);
                ahi[kk] = *reinterpret_cast<const uint4 *>(xa + 16);
                blo[kk] = *reinterpret_cast<const uint4 *>(xb);
                bhi[kk] = *reinterpret_cast<const uint4 *>(xb + 16);
            }
            __builtin_amdgcn_sched_barrier(0);         // keep the loads ahead of the MFMAs (the scheduler sinks them)
            if (EV_KNOCK & 128) {
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
                    const unsigned x = alo[kk].x ^ ahi[kk].y ^ blo[kk].z ^ bhi[kk].w ^ b0[kk].x ^ b1[kk].y;
                    acca[0] += __uint_as_float(x & 0x3F800000u);
                }
                continue;
            }
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                acca = mfma16<H>(alo[kk], b0[kk], acca);
                accb = mfma16<H>(blo[kk], b0[kk], accb);
                acca = mfma16<H>(ahi[kk], b1[kk], acca);
                accb = mfma16<H>(bhi[kk], b1[kk], accb);
            }
        }
        store_tile(acca, g0, row_end);
        store_tile(accb, g0 + 32, row_end);
    };
    // X path: rows [g0, g0 + 32), gathered rows read from X through the int32 table (events too large for the LDS, row
    // sets not grouped by event).  A plain loop over the offsets: this path only has to be right, and must not cost the
    // LDS path registers (unrolled, its 27 table entries and ballots did).
    auto tile_glb = [&](int g0, int row_end) {
        const int row = g0 + r;
        const bool live = row < row_end;
        const unsigned ra = (unsigned)(live ? row : row_end - 1);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = bj;
#pragma unroll 1
        for (int k = 0; k < K; ++k) {
            // one uniform base + a 32-bit element offset (the launcher checks K * R < 2^31)
            int nb = (k == identity_k) ? row : table[(unsigned)(mirror ? K - 1 - k : k) * (unsigned)R + ra];
            nb = live ? nb : -1;
            if (__ballot(nb >= 0) == 0ull) continue;
            const uint4 *xp = (const uint4 *)(X + (long long)(nb >= 0 ? nb : 0) * 32 + h * 16);
            const uint4 lo = keep_if(xp[0], nb >= 0), hi = keep_if(xp[1], nb >= 0);
            const uint4 *bp = sWb + k * 128 + h * 32 + r;
            acc = mfma16<H>(lo, bp[0], acc);
            acc = mfma16<H>(hi, bp[64], acc);
        }
        store_tile(acc, g0, row_end);
    };
    // 16-byte chunk c of the event's rows -> its place in sX (input row j is LDS row j + 1)
    auto put_chunk = [&](int c, uint4 val) {
        *reinterpret_cast<uint4 *>(sXb + ((c >> 2) + 1) * EV_STRIDE + ((c & 3) << 4)) = val;
    };

    // ---- pass 1: the events whose input rows fit the LDS, one event at a time
    uint4 xr[EV_ROWLOADS];
    bool fits = false;
    int oend = 0, nchunk = 0, g_first = 0;
    const uint4 *xsrc = nullptr;
    auto fetch_event = [&]() {            // descriptor in o0 / o1 / i0 / i1: issue the row loads and the first slot loads
        oend = o1 < Rv ? o1 : Rv;
        fits = structured && e < B && (i1 - i0) <= cap;
        nchunk = (fits && !(EV_KNOCK & 4)) ? (i1 - i0) * 4 : 0;
        xsrc = reinterpret_cast<const uint4 *>(X + (long long)i0 * 32);
        const int clast = nchunk > 0 ? nchunk - 1 : 0;       // unconditional, clamped: no branch per load
        // (every register below is DEFINED on every path: a value left over from the previous event would otherwise
        // count as live through the tile phase and cost its registers there)
        if (fits) {
#pragma unroll
            for (int b = 0; b < EV_ROWLOADS; ++b) {
                const int c = threadIdx.x + b * EV_THREADS;
                xr[b] = xsrc[c < clast ? c : clast];
            }
        } else {
#pragma unroll
            for (int b = 0; b < EV_ROWLOADS; ++b) xr[b] = uint4{0u, 0u, 0u, 0u};
        }
        g_first = o0 + 64 * wid;
        if (fits && g_first < oend) {
            load_slots(g_first, oend);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) ca[q] = cb[q] = uint4{0u, 0u, 0u, 0u};
        }
    };
    fetch_event();
#pragma unroll
    for (int b = 0; b < WSB; ++b) {
        const int u = threadIdx.x + b * EV_THREADS;
        uint4 pk;
        pk.x = wfs_pack2<H>(w[b][0], w[b][1]);
        pk.y = wfs_pack2<H>(w[b][2], w[b][3]);
        pk.z = wfs_pack2<H>(w[b][4], w[b][5]);
        pk.w = wfs_pack2<H>(w[b][6], w[b][7]);
        if (u < nfrag && !(EV_KNOCK & 8)) sWb[u] = pk;
    }
    if (threadIdx.x < 5) reinterpret_cast<uint4 *>(sXb)[threadIdx.x] = uint4{0u, 0u, 0u, 0u};      // the zero row
    while (structured && e < B) {
        if (fits) {
#pragma unroll
            for (int b = 0; b < EV_ROWLOADS; ++b) {
                const int c = threadIdx.x + b * EV_THREADS;
                if (c < nchunk) put_chunk(c, xr[b]);
            }
            // rows beyond the first batch (events of more than 1024 rows): one more round trip per batch
            for (int c = EV_ROWLOADS * EV_THREADS + threadIdx.x; c < nchunk; c += EV_THREADS) put_chunk(c, xsrc[c]);
        }
        __syncthreads();
        if (fits) {
            if (EV_KNOCK & 64) oend = oend < o0 + 256 ? oend : o0 + 256;      // timing: no event larger than 256 rows
            for (int rep = 0; rep < ((EV_KNOCK & 32) ? 2 : 1); ++rep)           // timing: the tile phase twice
            for (int g0 = g_first; g0 < oend; g0 += 64 * EV_WAVES) {
                if (g0 != g_first || rep) load_slots(g0, oend);
                tile_lds(g0, oend);
            }
        }
        e += gridDim.x;
        if (e >= B) break;
        __syncthreads();                       // every wave is done with the previous event's rows
        o0 = out_ev[e];
        o1 = out_ev[e + 1];
        i0 = in_ev[e];
        i1 = in_ev[e + 1];
        fetch_event();
    }
    __syncthreads();                           // the filters are in place (blocks without an event come straight here)
    // ---- pass 2, operands gathered from X: events too large for the LDS -- or, when a row set is not grouped by
    // event, 32-row tiles over the whole row range
    int e2 = structured ? (int)blockIdx.x : 0;
    while (true) {
        int lo, hi, t0, tstride;
        if (structured) {
            if (e2 >= B) break;
            lo = out_ev[e2];
            hi = out_ev[e2 + 1];
            hi = hi < Rv ? hi : Rv;
            const int nin = in_ev[e2 + 1] - in_ev[e2];
            e2 += gridDim.x;
            if (nin <= cap) continue;
            t0 = wid;
            tstride = EV_WAVES;
        } else {
            if (e2 > 0) break;
            e2 = 1;
            lo = 0;
            hi = Rv;
            t0 = blockIdx.x * EV_WAVES + wid;
            tstride = gridDim.x * EV_WAVES;
        }
        for (int g0 = lo + 32 * t0; g0 < hi; g0 += 32 * tstride) tile_glb(g0, hi);
    }
}

}  // namespace

extern "C" int wfs_event_conv_ok(int32_t K, int32_t Cx, int32_t Cw_in, int32_t Cw_out, int32_t dtype,
                                 int32_t batch_size) {
    return (dtype == WFS_BF16 || dtype == WFS_F16) && Cx == 32 && Cw_in == 32 && Cw_out == 32 && K >= 1 && K <= 27 &&
           batch_size >= 1;
}

extern "C" int wfs_slot_table(const int32_t *table, int32_t mirror, int32_t K, int32_t identity_k, int64_t R,
                              const int32_t *out_events, const int32_t *in_events, int32_t batch_size,
                              const int64_t *r_dev, void *slots, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(K >= 1 && K <= 32 && identity_k < K && batch_size >= 1, WFS_EINVAL, "wfs_slot_table covers K <= 32");
    if (R == 0) return WFS_OK;
    WFS_REQUIRE(table && out_events && in_events && slots, WFS_EINVAL, "NULL device pointer");
    WFS_REQUIRE(R < (1ll << 31), WFS_EINVAL, "R out of range");
    WfsTimerScope timer(WFS_TIMER_RULEBOOK, stream);
    const int nblk = batch_size < 2048 ? batch_size : 2048;
    k_slot_table<<<dim3((unsigned)nblk), dim3(256), 0, stream>>>(table, mirror, K, identity_k, R, (const long long *)r_dev,
                                                                out_events, in_events, batch_size, (uint4 *)slots);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

template <typename H>
static int launch_evconv(const int *table, int mirror, int K, int identity_k, long long R, const long long *r_dev,
                         const uint4 *ctab, int slots_mirror, const int *out_ev, const int *in_ev, int B, const H *X,
                         const float *W, int transpose_w, const float *bias, H *Y, hipStream_t stream) {
    // LDS: filters K * 2 KiB, then as many 80-byte rows as fit under 160 KiB (+ the zero row)
    const size_t budget = 160 * 1024 - 512;
    int cap = (int)((budget - (size_t)K * 2048) / EV_STRIDE) - 1;
    if (cap > 65534) cap = 65534;
    const size_t lds = (size_t)K * 2048 + (size_t)(cap + 1) * EV_STRIDE;
    long long nblk = B < 256 ? B : 256;
    // a row set that turns out not to be grouped by event is covered by the same grid, tile-parallel: give it the chip
    if (nblk < 256 && R > 32ll * EV_WAVES * nblk) {
        nblk = (R + 32 * EV_WAVES - 1) / (32 * EV_WAVES);
        if (nblk > 256) nblk = 256;
    }
    static bool attr[4] = {false, false, false, false};
    auto kern = slots_mirror ? (transpose_w ? k_evconv32<H, true, true> : k_evconv32<H, false, true>)
                             : (transpose_w ? k_evconv32<H, true, false> : k_evconv32<H, false, false>);
    bool *done = &attr[(transpose_w ? 1 : 0) + (slots_mirror ? 2 : 0)];
    if (!*done) {
        WFS_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget));
        *done = true;
    }
    kern<<<dim3((unsigned)nblk), dim3(EV_THREADS), lds, stream>>>(table, mirror, K, identity_k, R, r_dev, ctab, out_ev,
                                                                   in_ev, B, X, W, bias, Y, cap);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

extern "C" int wfs_event_conv(const int32_t *table, int32_t mirror, int32_t K, int32_t identity_k, int64_t R,
                              const void *slots, int32_t slots_mirror, const int32_t *out_events,
                              const int32_t *in_events, int32_t batch_size, const void *X, const float *W,
                              int32_t transpose_w, const float *bias, void *Y, int32_t dtype, const int64_t *r_dev,
                              void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(wfs_event_conv_ok(K, 32, 32, 32, dtype, batch_size), WFS_EINVAL,
                "wfs_event_conv covers 32 -> 32 channels, 16-bit rows, K <= 27");
    if (R == 0) return WFS_OK;
    WFS_REQUIRE(table && slots && out_events && in_events && X && W && Y, WFS_EINVAL, "NULL device pointer");
    WFS_REQUIRE(identity_k < K && (long long)K * R < (1ll << 31), WFS_EINVAL, "identity_k / K * R out of range");
    WFS_REQUIRE(!slots_mirror || K == 27, WFS_EINVAL, "mirrored slot records: K = 27 only");
    WfsTimerScope timer(WFS_TIMER_GATHER_CONV, stream);
    if (dtype == WFS_F16)
        return launch_evconv<wfs_f16>(table, mirror, K, identity_k, R, (const long long *)r_dev, (const uint4 *)slots,
                                      slots_mirror, out_events, in_events, batch_size, (const wfs_f16 *)X, W, transpose_w,
                                      bias, (wfs_f16 *)Y, stream);
    return launch_evconv<wfs_bf16>(table, mirror, K, identity_k, R, (const long long *)r_dev, (const uint4 *)slots,
                                   slots_mirror, out_events, in_events, batch_size, (const wfs_bf16 *)X, W, transpose_w,
                                   bias, (wfs_bf16 *)Y, stream);
}
