// evrulebook.hip -- EVENT-LOCAL rulebook construction (round 3).
//
// Replaces torch.ops.spconv.get_indice_pairs of spconv 1.2.1 (reference requirements.txt:15; call sites
// src/models/SPConvBlocks.py:75,134,498) for index sets that are grouped by event -- what the reference's collate_fn
// delivers (src/engineering/PSDDataModule.py:10-20) -- with ONE WORKGROUP PER EVENT: the rulebook key includes the batch
// index, so an event's sites can be looked up in a table that lives in LDS.  rulebook.hip's chip-wide
// form pays for a site grid over the whole batch in HBM (18.5 MB cleared per build at the PSD batch, one atomic or one
// dependent global read per candidate: 36.7 MB of traffic for a 6.3 MB result); here the only HBM traffic is the
// coordinates in and the tables out.
//
//   k_ev_subm   SubM: the event's sites go into an LDS hash (key = row-major site, value = local row, duplicates: the last
//               row wins = atomicMax, as A.3), every (row, offset) candidate is one probe.  Output: nbr_out [K, N] exactly as
//               rulebook.hip writes it (bit-identical), and optionally the per-event slot records evconv.hip consumes.
//   (the regular / strided counterpart, k_ev_conv, is an experiment that did not beat rulebook.hip's build -- one workgroup
//   per event is bound by the largest event: tools/exp/event_local/, profiles/r03_event_local_*)
#include <stdlib.h>

#include "wfs_common.h"

namespace {

struct EGeo {
    int ndim, K;
    int spatial[4], out_shape[4], ksize[4], stride[4], padding[4], dilation[4];
};

// timing knock-outs (results wrong by construction): 1 no lookups / stores, 2 no table stores, 4 no inserts, 8 no probes
#ifndef ER_KNOCK
#define ER_KNOCK 0
#endif
constexpr int EV_FLAG_BLOCKS = WFS_EVENT_FLAG_WORDS;      // k_event_offsets runs this many blocks, one flag word each
constexpr int ER_THREADS = 512;

__device__ __forceinline__ long long valid_rows(long long R, const long long *r_dev) {
    long long v = r_dev ? *r_dev : R;
    return v < R ? v : R;
}

__device__ __forceinline__ bool ev_structured(const int *ev, int B) {
    const int fl = ev[B + 1 + (threadIdx.x & 63)];
    return __ballot(fl != 0) == 0ull;
}

// offsets of kernel position k (last dim fastest), per dim
__device__ __forceinline__ void offset_digits(const EGeo &g, int k, int *off) {
    int rem = k;
#pragma unroll
    for (int d = 3; d >= 0; --d) {
        off[d] = 0;
        if (d >= g.ndim) continue;
        off[d] = rem % g.ksize[d];
        rem /= g.ksize[d];
    }
}

// ------------------------------------------------------------------------------------------ event offsets
// off[e] = first row of event e (e = 0 .. B; off[B] = number of valid rows); off[B + 1 + blk] = 1 if block blk saw a
// batch index out of [0, B) or smaller than its predecessor's.  Every word is written by every launch (no clearing).
__global__ void __launch_bounds__(256) k_event_offsets(const int *__restrict__ idx, long long N, int cols, int B,
                                                       const long long *__restrict__ n_dev, int *__restrict__ off) {
    const long long Nv = valid_rows(N, n_dev);
    int bad = 0;
    if (Nv == 0) {
        for (int e = blockIdx.x * 256 + threadIdx.x; e <= B; e += gridDim.x * 256) off[e] = 0;
    }
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < Nv; j += (long long)gridDim.x * 256) {
        const int b = idx[j * cols];
        const int bp = j > 0 ? idx[(j - 1) * cols] : -1;
        const bool ok = b >= 0 && b < B && b >= bp && bp >= -1 && bp < B;
        bad |= ok ? 0 : 1;
        if (ok) {
            for (int e = bp + 1; e <= b; ++e) off[e] = (int)j;
            if (j == Nv - 1)
                for (int e = b + 1; e <= B; ++e) off[e] = (int)Nv;
        }
    }
    bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) off[B + 1 + blockIdx.x] = bad;
}

// ------------------------------------------------------------------------------------------ SubM
// Site table of ONE event in LDS, direct addressing in two levels -- the detector's shape: a few active cells (leading
// dims: the PMT grid), each a dense run of samples (last dim: time):
//     cell_slot[cell]            -1 or the index of the cell's sample array      (L = prod(leading dims) entries)
//     pool[slot * T + t]         0 or 1 + local row                               (uint16; `nslot` arrays of T samples)
// A lookup is two dependent LDS reads, no probing, no key compare; the kl candidates of a row that differ only in the
// last dim's offset share the first.  (An open-addressing hash was measured first: at ~110 instructions per candidate --
// probe loops under divergence -- the largest event kept one CU busy for 30 us.)
// Capacity: L <= ER_MAXCELLS, active cells * T * 2 bytes <= ER_POOL bytes, rows <= 65534 per event; beyond that
// flags[0] is set, the affected rows' table entries all say "no neighbour" (never left unwritten) and the caller takes
// rulebook.hip's build.  Duplicate coordinates are DETECTED (flags[1]; a row that
// does not read its own index back), not resolved: "the last row wins" (A.3) is then the caller's, i.e. rulebook.hip's.
// flags: int32 [3 * blocks] (wfs_event_rulebook_flag_ints), every word written by every launch: [0 .. blocks) not grouped
// by event / capacity, [blocks .. 2 blocks) duplicates, [2 blocks .. 3 blocks) an index outside the spatial shape.
constexpr int ER_MAXCELLS = 2048;
constexpr int ER_POOL = 64 * 1024;

// per leading-offset index q (host-computed): cell difference and the packed offset digits of the leading dims
struct EQTab {
    int dcell[32];
    unsigned off[32];
};

template <int KL>
__global__ void __launch_bounds__(ER_THREADS) k_ev_subm(EGeo g, EQTab qt, int Q, int L, int split, int pool_bytes,
                                                        const int *__restrict__ idx,
                                                        long long N, const long long *__restrict__ n_dev,
                                                        const int *__restrict__ ev, int B, int *__restrict__ nbr_out,
                                                        uint4 *__restrict__ slots, int *__restrict__ flags) {
    __shared__ int cell_slot[ER_MAXCELLS];
    extern __shared__ __attribute__((aligned(16))) unsigned short pool[];      // pool_bytes
    __shared__ int2 sQt[32];
    __shared__ int sCount[ER_THREADS / 64 + 1];
    const int Nv = (int)valid_rows(N, n_dev);
    // flags: one word per workgroup and kind, flags[kind * gridDim.x + block]; set, never cleared, by a launch (see the end)
    int f_fail = 0, f_dup = 0, f_range = 0;
    const bool structured = ev_structured(ev, B);
    if (!structured) f_fail = 1;
    if (threadIdx.x < 32) sQt[threadIdx.x] = int2{qt.dcell[threadIdx.x], (int)qt.off[threadIdx.x]};
    const int cols = g.ndim + 1, last = g.ndim - 1;
    const int T = g.spatial[last];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // A build that fails (flags below) must not leave table entries unwritten: the table lives in recycled memory and the
    // consumers gather through it until the runner reads the flags.  "No neighbour" everywhere is benign (the products
    // add the centre offset themselves).
    auto fill_none = [&](long long j_lo, long long j_hi) {
        for (int k = 0; k < g.K; ++k)
            for (long long j = j_lo + threadIdx.x; j < j_hi; j += ER_THREADS) nbr_out[(long long)k * N + j] = -1;
    };
    if (!structured) {
        // the event table is meaningless: the workgroups share ALL the valid rows out evenly
        const long long per = (Nv + gridDim.x - 1) / gridDim.x;
        const long long lo = (long long)blockIdx.x * per;
        fill_none(lo, lo + per < Nv ? lo + per : Nv);
    }
    // `split` workgroups per event: each builds the event's table and serves its share of the rows
    for (int item = blockIdx.x; structured && item < B * split; item += gridDim.x) {
        const int e = item / split, part = item % split;
        const int o0 = ev[e];
        int o1 = ev[e + 1];
        o1 = o1 < Nv ? o1 : Nv;
        const int n = o1 - o0;
        if (n <= 0) continue;
        const int j_lo = (int)((long long)n * part / split), j_hi = (int)((long long)n * (part + 1) / split);
        if (n > 65534) {
            f_fail = 1;
            fill_none(o0 + j_lo, o0 + j_hi);
            continue;
        }
        // a row's leading coordinates -> cell (-1: outside the shape), last coordinate -> t
        auto load_row = [&](int j, int *x, int &t) -> int {
            const int *row = idx + (long long)(o0 + j) * cols;
            bool ok = true;
            int cell = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                x[d] = d < last ? row[1 + d] : 0;
                if (d < last) {
                    ok = ok && x[d] >= 0 && x[d] < g.spatial[d];
                    cell = cell * g.spatial[d] + x[d];
                }
            }
            t = row[1 + last];
            ok = ok && t >= 0 && t < T;
            return ok ? cell : -1;
        };
        __syncthreads();                              // the previous event's lookups are done
        for (int c = threadIdx.x; c < L; c += ER_THREADS) cell_slot[c] = 0;
        __syncthreads();
        // pass A: which cells are active
#pragma unroll 1
        for (int j = threadIdx.x; j < n; j += ER_THREADS) {
            int x[3], t;
            const int cell = load_row(j, x, t);
            if (cell < 0)
                f_range = 1;
            else
                cell_slot[cell] = 1;
        }
        __syncthreads();
        // pass B: slot = number of active cells in front (block scan over the L cells, 512 at a time)
        int carry = 0;
        for (int c0 = 0; c0 < L; c0 += ER_THREADS) {
            const int c = c0 + threadIdx.x;
            const int act = c < L ? cell_slot[c] : 0;
            const unsigned long long bal = __ballot(act != 0);
            if (lane == 0) sCount[wid] = __popcll(bal);
            __syncthreads();
            int base = carry, tot = 0;
#pragma unroll
            for (int w = 0; w < ER_THREADS / 64; ++w) {
                const int cw = sCount[w];
                if (w < wid) base += cw;
                tot += cw;
            }
            if (c < L) cell_slot[c] = act ? base + __popcll(bal & ((1ull << lane) - 1ull)) : -1;
            carry += tot;
            __syncthreads();
        }
        const int nslot = carry;
        if ((long long)nslot * T * 2 > pool_bytes) {
            f_fail = 1;
            fill_none(o0 + j_lo, o0 + j_hi);
            continue;
        }
        // pass C: clear the active cells' sample arrays (dwords)
        {
            unsigned *p32 = reinterpret_cast<unsigned *>(pool);
            const int nd = (nslot * T + 1) >> 1;
            for (int i = threadIdx.x; i < nd; i += ER_THREADS) p32[i] = 0u;
        }
        __syncthreads();
        // pass D: rows into the table
#pragma unroll 1
        for (int j = threadIdx.x; j < n; j += ER_THREADS) {
            int x[3], t;
            const int cell = load_row(j, x, t);
            if (cell >= 0) pool[cell_slot[cell] * T + t] = (unsigned short)(j + 1);
        }
        __syncthreads();
        // pass E: every row (of this workgroup's share) looks its K candidates up and writes its table column entries
#pragma unroll 1
        for (int j = j_lo + threadIdx.x; j < j_hi; j += ER_THREADS) {
            int x[3], t;
            const int cell = load_row(j, x, t);
            unsigned rec[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) rec[q] = 0u;
            // leading dims: offset o of dim d keeps the candidate inside the shape <=> bit o of vm[d]
            unsigned vm[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                vm[d] = d < last ? 0u : 1u;
                if (d < last && cell >= 0)
                    for (int o = 0; o < g.ksize[d]; ++o) {
                        const int c = x[d] + g.padding[d] - o * g.dilation[d];
                        vm[d] |= (c >= 0 && c < g.out_shape[d]) ? (1u << o) : 0u;
                    }
            }
            if (cell >= 0 && pool[cell_slot[cell] * T + t] != (unsigned short)(j + 1)) f_dup = 1;         // duplicate site
#pragma unroll
            for (int q = 0; q < 32 / KL; ++q) {
                if (q >= Q) break;
                const int2 qd = sQt[q];
                const unsigned op = (unsigned)qd.y;
                const unsigned okb = (vm[0] >> (op & 255u)) & (vm[1] >> ((op >> 8) & 255u)) & (vm[2] >> ((op >> 16) & 255u)) & 1u;
                const int cs = (cell >= 0 && okb) ? cell_slot[cell + qd.x] : -1;
                const int pbase = cs * T;
#pragma unroll
                for (int o = 0; o < KL; ++o) {
                    const int k = q * KL + o;
                    const int tt = t + g.padding[last] - o * g.dilation[last];
                    const bool in = cs >= 0 && tt >= 0 && tt < T;
                    const unsigned v = pool[in ? pbase + tt : 0];
                    const unsigned sv = in ? v : 0u;
                    int *col = nbr_out + (long long)k * N;          // uniform base + 32-bit row offset
                    if (!(ER_KNOCK & 2) || sv == 0x12345u) col[(unsigned)(o0 + j)] = sv ? o0 + (int)sv - 1 : -1;
                    rec[k >> 1] |= (k & 1) ? (sv << 16) : sv;
                }
            }
            if (slots && (!(ER_KNOCK & 2) || rec[3] == 0x12345678u)) {
                uint4 *dst = slots + (long long)(o0 + j) * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = uint4{rec[4 * q], rec[4 * q + 1], rec[4 * q + 2], rec[4 * q + 3]};
            }
        }
    }
    f_fail = __syncthreads_or(f_fail);
    f_dup = __syncthreads_or(f_dup);
    f_range = __syncthreads_or(f_range);
    if (threadIdx.x == 0) {
        // STICKY: a flag is only ever set here; whoever reads them clears them (a captured step is checked every so many
        // replays: a failure of any replay in between must still be there)
        if (f_fail) flags[blockIdx.x] = 1;
        if (f_dup) flags[gridDim.x + blockIdx.x] = 1;
        if (f_range) flags[2 * gridDim.x + blockIdx.x] = 1;
    }
}

EGeo make_egeo(const wfs_geometry *g) {
    EGeo G;
    G.ndim = g->ndim;
    G.K = g->K;
    for (int i = 0; i < 4; ++i) {
        G.spatial[i] = g->spatial[i];
        G.out_shape[i] = g->out_shape[i];
        G.ksize[i] = g->ksize[i];
        G.stride[i] = g->stride[i];
        G.padding[i] = g->padding[i];
        G.dilation[i] = g->dilation[i];
    }
    return G;
}

// SubM geometry (out_shape = spatial): per leading-offset index q the cell difference and the packed digits
EQTab make_qtab_subm(const wfs_geometry *g, int *Q, int *L) {
    EQTab t;
    const int last = g->ndim - 1;
    int q_count = 1, cells = 1;
    for (int d = 0; d < last; ++d) {
        q_count *= g->ksize[d];
        cells *= g->spatial[d];
    }
    *Q = q_count;
    *L = cells;
    for (int q = 0; q < 32; ++q) {
        t.dcell[q] = 0;
        t.off[q] = 0;
        if (q >= q_count) continue;
        int rem = q, off[4] = {0, 0, 0, 0};
        for (int d = last - 1; d >= 0; --d) {
            off[d] = rem % g->ksize[d];
            rem /= g->ksize[d];
        }
        long long dc = 0;
        for (int d = 0; d < last; ++d) dc = dc * g->out_shape[d] + (g->padding[d] - off[d] * g->dilation[d]);
        t.dcell[q] = (int)dc;
        t.off[q] = (unsigned)off[0] | ((unsigned)off[1] << 8) | ((unsigned)off[2] << 16);
    }
    return t;
}

}  // namespace

extern "C" int wfs_event_rulebook_ok(const wfs_geometry *g) {
    if (!g || g->K < 1 || g->K > 32 || g->transposed || g->ndim < 1 || g->ndim > WFS_MAX_DIM) return 0;
    const int last = g->ndim - 1;
    if (g->ksize[last] > 3) return 0;                 // instantiated for 1, 2, 3 offsets along the last dim
    long long cells = 1;
    for (int d = 0; d < last; ++d) {
        if (g->ksize[d] > 31) return 0;
        cells *= g->subm ? g->spatial[d] : g->out_shape[d];
    }
    if (cells > ER_MAXCELLS) return 0;
    const long long T = g->subm ? g->spatial[last] : g->out_shape[last];
    return T * 2 <= ER_POOL;                          // at least one cell's sample array fits
}

static int subm_blocks(int B) {
    const long long items = (long long)B * 2;
    return (int)(items < 2048 ? items : 2048);
}

extern "C" size_t wfs_event_rulebook_flag_ints(int32_t batch_size) { return 3 * (size_t)subm_blocks(batch_size > 0 ? batch_size : 1); }

extern "C" int wfs_event_rulebook_subm(const wfs_geometry *g, const int32_t *indices, int64_t N, const int64_t *n_dev,
                                       const int32_t *events, int32_t *nbr_out, void *slots, int32_t *flags,
                                       void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(g && g->subm && wfs_event_rulebook_ok(g), WFS_EINVAL, "wfs_event_rulebook_subm: SubM geometry, K <= 32");
    WFS_REQUIRE(N >= 0 && (long long)g->K * N < (1ll << 31), WFS_EINVAL, "N out of range");
    if (N == 0) return WFS_OK;
    WFS_REQUIRE(indices && events && nbr_out && flags, WFS_EINVAL, "NULL device pointer");
    WfsTimerScope timer(WFS_TIMER_RULEBOOK, stream);
    const int B = g->batch_size;
    // sample arrays for every cell of an event when that fits 64 KiB, else for as many active cells as 64 KiB hold (128
    // at 256 samples; a PSD event has <= ~20): an event with more is flagged, never mis-built
    int Q = 1, L = 1;
    const EQTab qt = make_qtab_subm(g, &Q, &L);
    const int T = g->spatial[g->ndim - 1];
    long long pool_ll = (long long)L * T * 2;
    pool_ll = pool_ll < 1024 ? 1024 : (pool_ll > ER_POOL ? ER_POOL : pool_ll);
    const int pool_bytes = (int)((pool_ll + 15) / 16 * 16);
    const int split = 2;          // measured at the PSD batch: 1 -> 15.6 us, 2 -> 12.7 us, 4 -> 18.9 us
    const int nblk = subm_blocks(B);
    const EGeo G = make_egeo(g);
    const dim3 grid((unsigned)nblk), block(ER_THREADS);
    const int kl = g->ksize[g->ndim - 1];
#define WFS_EVS(KL)                                                                                                    \
    k_ev_subm<KL><<<grid, block, pool_bytes, stream>>>(G, qt, Q, L, split, pool_bytes, indices, N,                     \
                                              (const long long *)n_dev, events, B, nbr_out,                            \
                                              (uint4 *)slots, flags)
    if (kl == 1) WFS_EVS(1); else if (kl == 2) WFS_EVS(2); else WFS_EVS(3);
#undef WFS_EVS
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

extern "C" size_t wfs_event_offsets_ints(int32_t batch_size) {
    return (size_t)(batch_size > 0 ? batch_size : 0) + 1 + EV_FLAG_BLOCKS;
}

extern "C" int wfs_event_offsets(const int32_t *indices, int64_t N, int32_t ndim, int32_t batch_size,
                                 const int64_t *n_dev, int32_t *offsets, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(offsets && (indices || N == 0), WFS_EINVAL, "NULL device pointer");
    WFS_REQUIRE(ndim >= 1 && ndim <= WFS_MAX_DIM && batch_size >= 1 && N >= 0 && N < (1ll << 31), WFS_EINVAL,
                "bad shape");
    k_event_offsets<<<dim3(EV_FLAG_BLOCKS), dim3(256), 0, stream>>>(indices, N, ndim + 1, batch_size,
                                                                    (const long long *)n_dev, offsets);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}
