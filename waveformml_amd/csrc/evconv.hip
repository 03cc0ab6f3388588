// evconv.hip -- EVENT-LOCAL rulebook construction for REGULAR (strided) sparse convolutions, one launch (round 4).
//
// Replaces torch.ops.spconv.get_indice_pairs(subm=False) of spconv 1.2.1 (reference requirements.txt:15; call sites
// src/models/SPConvBlocks.py:75,498) for index sets that are grouped by event -- what the reference's collate_fn
// delivers (src/engineering/PSDDataModule.py:10-20) and what every regular conv built here produces (first-seen
// numbering of an event-grouped input is event-grouped).  The algorithm matched bit for bit is SURVEY.md A.3 as
// restated in oracle/spconv_ref.c:208-276: output sites are numbered in the order in which the sequential walk over
// (input row, kernel offset) first reaches them.
//
// rulebook.hip's chip-wide form runs six launches over dense -1-padded tables (k_ws_init -> insert -> first flags ->
// row bases -> assign -> finalize: 61 us and 74 MB of traffic for a 6 MB result at the PSD batch).  Here ONE WORKGROUP
// OWNS ONE EVENT, whose output grid (<= EC_MAXCELLS cells) lives in LDS:
//   1  every (row, offset) candidate does ds_min(ticket[cell], local row * 32 + offset)            [LDS atomics]
//   2  a candidate that reads its own ticket back is the FIRST to reach its site; a row's first flags are a 32-bit
//      mask; block-wide exclusive scan over the rows in order -> the site's id WITHIN the event           [LDS]
//   3  the event's site count is published (one 64-bit word, tagged with the launch's epoch) and the counts of all
//      events in front are read back: ids of event e start at the sum of the counts of events < e (events are
//      numbered in row order, so this IS first-seen order)            [one global store + one round of loads]
//   4  every candidate reads its site's id from LDS and the tables are written ONCE: out_indices, the by-input
//      table (packed [K / kl, N] when at most one offset along the last dimension can reach an output cell, i.e.
//      kernel <= stride there: 9 instead of 27 rows at the PSD geometry), the by-output table [K, M], the event
//      offsets of the OUTPUT set (the next strided layer starts from them) and, on request, the cell -> row map
//      that dense() of the output uses.
// No site grid in HBM, no clearing launch, no global read-modify-write; HBM traffic = coordinates in, tables out.
#include <stdlib.h>

#include "wfs_common.h"

namespace {

// timing knock-outs (results wrong by construction; tools/exp/eck<bits>/ via `make knock_ec`): 1 no look-back (ids start at
// 0 in every event), 2 no by-output table, 4 no table / coordinate stores at all, 8 no tickets / first flags
#ifndef EC_KNOCK
#define EC_KNOCK 0
#endif
constexpr int EC_THREADS = 1024;
constexpr int EC_WAVES = EC_THREADS / 64;
constexpr int EC_MAXCELLS = 16384;            // output cells of one event: 4 B ticket + 2 B id each in LDS
constexpr int EC_FLAG_WORDS = WFS_EVENT_FLAG_WORDS;
constexpr unsigned EC_SPIN_LIMIT = 1u << 22;  // polls of a predecessor's count before the launch gives up (flagged)

struct ECGeo {
    int ndim, K, Kq, kl, sl, pl, dl, out_last, cells_e;
    int spatial[4];
    // leading dims (all but the last; unused ones: ksize 1, out 1, mult 0)
    int lks[3], ls[3], lp[3], ld[3], lout[3], lmult[3];
    unsigned lmagic[3], last_magic;          // floor(t / s) == (t * magic) >> 16 for the t that occur (host-verified)
    unsigned qdig[32];                       // leading-offset digits of q, 8 bits per dim
};

__device__ __forceinline__ long long valid_rows(long long R, const long long *r_dev) {
    long long v = r_dev ? *r_dev : R;
    return v < R ? v : R;
}

// a row's coordinates, digested: per leading dim the offsets that reach an output cell (bit mask) and that cell's
// coordinate (8 bits per offset); for the last dim the coordinate itself
struct RowC {
    bool ok;
    unsigned vm[3], pk[3];
    int tl;                 // x_last + padding_last
};

__device__ __forceinline__ RowC digest_row(const ECGeo &g, const int *__restrict__ row) {
    RowC r;
    r.ok = true;
    const int last = g.ndim - 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        r.vm[d] = 1u;
        r.pk[d] = 0u;
        if (d < last) {
            const int x = row[1 + d];
            r.ok = r.ok && x >= 0 && x < g.spatial[d];
            unsigned vm = 0u, pk = 0u;
            for (int o = 0; o < g.lks[d]; ++o) {
                const int t = x + g.lp[d] - o * g.ld[d];
                const int oc = (int)(((unsigned)(t < 0 ? 0 : t) * g.lmagic[d]) >> 16);
                const bool v = t >= 0 && oc * g.ls[d] == t && oc < g.lout[d];
                vm |= v ? (1u << o) : 0u;
                pk |= (unsigned)(v ? oc : 0) << (8 * o);
            }
            r.vm[d] = vm;
            r.pk[d] = pk;
        }
    }
    const int xl = row[1 + last];
    r.ok = r.ok && xl >= 0 && xl < g.spatial[last];
    r.tl = xl + g.pl;
    return r;
}

// calls f(q, o, cell) for every candidate of the row that reaches an output cell, in increasing k = q * kl + o
template <bool PACKED, typename F>
__device__ __forceinline__ void each_candidate(const ECGeo &g, const RowC &r, F f) {
    if (!r.ok) return;
    int o_star = 0, oc_star = 0;
    bool l_ok = true;
    if (PACKED) {
        // kernel <= stride along the last dim (dilation 1): only offset (x + p) mod s can divide
        oc_star = (int)(((unsigned)r.tl * g.last_magic) >> 16);
        o_star = r.tl - oc_star * g.sl;
        l_ok = o_star < g.kl && oc_star < g.out_last;
    }
#pragma unroll 1
    for (int q = 0; q < g.Kq; ++q) {
        const unsigned dg = g.qdig[q];
        const unsigned o0 = dg & 255u, o1 = (dg >> 8) & 255u, o2 = (dg >> 16) & 255u;
        if (!((r.vm[0] >> o0) & (r.vm[1] >> o1) & (r.vm[2] >> o2) & 1u)) continue;
        const int lead = (int)((r.pk[0] >> (8 * o0)) & 255u) * g.lmult[0] + (int)((r.pk[1] >> (8 * o1)) & 255u) * g.lmult[1] +
                         (int)((r.pk[2] >> (8 * o2)) & 255u) * g.lmult[2];
        if (PACKED) {
            if (l_ok) f(q, o_star, lead * g.out_last + oc_star);
        } else {
            for (int o = 0; o < g.kl; ++o) {
                const int t = r.tl - o * g.dl;
                if (t < 0) continue;
                const int oc = (int)(((unsigned)t * g.last_magic) >> 16);
                if (oc * g.sl == t && oc < g.out_last) f(q, o, lead * g.out_last + oc);
            }
        }
    }
}

// block-wide exclusive scan of one int per thread (threads in order); *total = the block's sum
__device__ __forceinline__ int ec_block_scan(int v, int *sWave, int *total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int n = __shfl_up(inc, d, 64);
        if (lane >= d) inc += n;
    }
    __syncthreads();                        // the previous use of sWave is over
    if (lane == 63) sWave[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < EC_WAVES; ++w) {
        const int s = sWave[w];
        base += w < wid ? s : 0;
        tot += s;
    }
    *total = tot;
    return base + inc - v;
}

template <bool PACKED>
__global__ void __launch_bounds__(EC_THREADS) k_ev_conv(ECGeo g, const int *__restrict__ idx, long long N,
                                                        const long long *__restrict__ n_dev,
                                                        const int *__restrict__ ev_in, int B, long long M_cap,
                                                        int *__restrict__ out_idx, long long *__restrict__ m_dev,
                                                        int *__restrict__ ev_out, int *__restrict__ nbr_out,
                                                        int *__restrict__ nbr_in, int *__restrict__ cell_row,
                                                        int *__restrict__ overflow, int *__restrict__ flags,
                                                        unsigned long long *__restrict__ pub,
                                                        unsigned *__restrict__ state) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ec_lds[];
    unsigned *ticket = reinterpret_cast<unsigned *>(ec_lds);                                  // [cells_e]
    unsigned short *cellid = reinterpret_cast<unsigned short *>(ec_lds + (size_t)g.cells_e * 4);   // [cells_e]
    __shared__ int sWave[EC_WAVES];
    __shared__ long long sSum[EC_WAVES];
    const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int cols = g.ndim + 1, last = g.ndim - 1;
    const unsigned tag = state[0] + 1u;                 // this launch's epoch: bumped by the LAST event's workgroup
    const long long Nv = valid_rows(N, n_dev);
    // the event table's flag words: any != 0 <=> the batch column is not grouped by event
    const bool structured = __ballot(ev_in[B + 1 + lane] != 0) == 0ull;
    int r0 = ev_in[e], r1 = ev_in[e + 1];
    r1 = r1 < (int)Nv ? r1 : (int)Nv;
    const int n = structured && r1 > r0 ? r1 - r0 : 0;
    int f_range = 0;

    for (int c = tid; c < g.cells_e; c += EC_THREADS) ticket[c] = 0xFFFFFFFFu;
    // this thread's first row stays in registers through all phases; events beyond EC_THREADS rows reload
    RowC mine;
    mine.ok = false;
    if (tid < n) {
        mine = digest_row(g, idx + (long long)(r0 + tid) * cols);
        if (!mine.ok) f_range = 1;
    }
    __syncthreads();
    // ---- 1: tickets
    for (int jl = tid; jl < n && !(EC_KNOCK & 8); jl += EC_THREADS) {
        const RowC r = jl == tid ? mine : digest_row(g, idx + (long long)(r0 + jl) * cols);
        if (jl != tid && !r.ok) f_range = 1;
        each_candidate<PACKED>(g, r, [&](int q, int o, int cell) {
            atomicMin(&ticket[cell], (unsigned)jl * 32u + (unsigned)(q * g.kl + o));
        });
    }
    __syncthreads();
    // ---- 2: first flags -> ids within the event
    int carry = 0;
    for (int j0 = 0; j0 < n && !(EC_KNOCK & 8); j0 += EC_THREADS) {
        const int jl = j0 + tid;
        RowC r;
        r.ok = false;
        if (jl < n) r = jl == tid ? mine : digest_row(g, idx + (long long)(r0 + jl) * cols);
        unsigned mask = 0u;
        each_candidate<PACKED>(g, r, [&](int q, int o, int cell) {
            const int k = q * g.kl + o;
            mask |= ticket[cell] == (unsigned)jl * 32u + (unsigned)k ? (1u << k) : 0u;
        });
        int tot;
        const int rowbase = carry + ec_block_scan(__popc(mask), sWave, &tot);
        each_candidate<PACKED>(g, r, [&](int q, int o, int cell) {
            const int k = q * g.kl + o;
            if ((mask >> k) & 1u) cellid[cell] = (unsigned short)(rowbase + __popc(mask & ((1u << k) - 1u)));
        });
        carry += tot;
    }
    const int M_e = carry;
    // ---- 3: publish the count, read the counts in front
    if (tid == 0)
        __hip_atomic_store(&pub[e], ((unsigned long long)tag << 32) | (unsigned long long)(unsigned)M_e, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    long long part = 0;
    int timed_out = 0;
    for (int p = tid; p < e && !(EC_KNOCK & 1); p += EC_THREADS) {
        unsigned long long w = __hip_atomic_load(&pub[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((unsigned)(w >> 32) != tag && spins < EC_SPIN_LIMIT) {
            __builtin_amdgcn_s_sleep(2);
            w = __hip_atomic_load(&pub[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ++spins;
        }
        if ((unsigned)(w >> 32) != tag) timed_out = 1;
        part += (long long)(unsigned)(w & 0xFFFFFFFFull);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    if (lane == 0) sSum[wid] = part;
    __syncthreads();                                    // also: cellid is complete
    long long base = 0;
#pragma unroll
    for (int w = 0; w < EC_WAVES; ++w) base += sSum[w];
    const long long m_total = base + M_e;               // meaningful in the last event's workgroup
    const long long m_room = base < M_cap ? M_cap - base : 0;        // ids of this event below m_room exist
    // ---- 4: the by-output table's rows of this event start as "no input" ...
    if (EC_KNOCK & 2) nbr_in = nullptr;
    if (nbr_in && !(EC_KNOCK & 4)) {
        const int mw = (long long)M_e < m_room ? M_e : (int)m_room;
        for (int k = wid; k < g.K; k += EC_WAVES) {
            int *dst = nbr_in + (long long)k * M_cap + base;
            for (int i = lane; i < mw; i += 64) dst[i] = -1;
        }
        __syncthreads();                                // ... before the candidates below fill theirs in
    }
    for (int jl = tid; jl < n && !(EC_KNOCK & 4); jl += EC_THREADS) {
        const RowC r = jl == tid ? mine : digest_row(g, idx + (long long)(r0 + jl) * cols);
        const long long j = (long long)r0 + jl;
        unsigned done = 0u;                             // offsets (PACKED: leading-offset rows) already written
        each_candidate<PACKED>(g, r, [&](int q, int o, int cell) {
            const int k = q * g.kl + o;
            const int il = (int)cellid[cell];
            const bool exists = il < m_room;
            const int id = exists ? (int)(base + il) : -1;
            if (PACKED) {
                nbr_out[(long long)q * N + j] = exists ? ((id << 3) | o) : -1;
                done |= 1u << q;
            } else {
                nbr_out[(long long)k * N + j] = id;
                done |= 1u << k;
            }
            if (!exists) return;
            if (nbr_in) nbr_in[(long long)k * M_cap + id] = (int)j;
            if (ticket[cell] == (unsigned)jl * 32u + (unsigned)k) {
                // first to reach the site: its coordinates
                int *o = out_idx + (long long)id * cols;
                int rem = cell;
                o[cols - 1] = rem % g.out_last;
                rem /= g.out_last;
#pragma unroll
                for (int d = 2; d >= 0; --d)
                    if (d < last) {
                        o[1 + d] = rem % g.lout[d];
                        rem /= g.lout[d];
                    }
                o[0] = e;
            }
        });
        const int nrows = PACKED ? g.Kq : g.K;
        for (int t = 0; t < nrows; ++t)
            if (!((done >> t) & 1u)) nbr_out[(long long)t * N + j] = -1;
    }
    if (!structured) {
        // the event table is meaningless: no row has an output (the consumers of the by-input table are bounded by the
        // INPUT row count, so every row they can reach must hold "none"); the workgroups share the rows out evenly
        const long long per = (Nv + B - 1) / B;
        const long long j_lo = (long long)e * per, j_hi = j_lo + per < Nv ? j_lo + per : Nv;
        const int nrows = PACKED ? g.Kq : g.K;
        for (int t = 0; t < nrows; ++t)
            for (long long j = j_lo + tid; j < j_hi; j += EC_THREADS) nbr_out[(long long)t * N + j] = -1;
    }
    if (cell_row) {
        int *dst = cell_row + (long long)e * g.cells_e;
        for (int c = tid; c < g.cells_e; c += EC_THREADS) {
            const int il = (int)cellid[c];
            dst[c] = (n > 0 && ticket[c] != 0xFFFFFFFFu && il < m_room) ? (int)(base + il) : -1;
        }
    }
    f_range = __syncthreads_or(f_range);
    timed_out = __syncthreads_or(timed_out);
    if (tid == 0) {
        ev_out[e] = (int)(base < M_cap ? base : M_cap);
        // STICKY: only ever set here (a captured step is checked every so many replays)
        if (!structured || timed_out) flags[0] = 1;
        if (f_range) flags[2] = 1;
        if (e == B - 1) {
            const long long m = !structured ? 0 : (m_total < M_cap ? m_total : M_cap);
            *m_dev = m;
            ev_out[B] = (int)m;
            if (overflow && m_total > M_cap) *overflow = 1;
            state[0] = tag;                             // every workgroup has read the old epoch (they all published)
        }
    }
    if (e == 0 && tid < EC_FLAG_WORDS) ev_out[B + 1 + tid] = ev_in[B + 1 + tid];
}

// expands a packed by-input table [K / kl, R] (entry = row << 3 | offset along the last kernel dim, or -1) to the
// dense [K, R] form
__global__ void __launch_bounds__(256) k_unpack_table(const int *__restrict__ packed, int Kq, int kl, long long R,
                                                      const long long *__restrict__ r_dev, int *__restrict__ dense) {
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    const int q = blockIdx.y;
    if (j >= valid_rows(R, r_dev) || q >= Kq) return;
    const int ev = packed[(long long)q * R + j];
    for (int o = 0; o < kl; ++o) dense[((long long)q * kl + o) * R + j] = (ev >= 0 && (ev & 7) == o) ? (ev >> 3) : -1;
}

bool magic_ok(int s, int tmax, unsigned *magic) {
    if (s < 1 || s > 256 || tmax < 0 || tmax >= (1 << 15)) return false;
    const unsigned m = (65536u + (unsigned)s - 1u) / (unsigned)s;
    for (int t = 0; t <= tmax; ++t)
        if ((int)(((unsigned)t * m) >> 16) != t / s) return false;
    *magic = m;
    return true;
}

bool make_ecgeo(const wfs_geometry *g, ECGeo *G) {
    if (!g || g->subm || g->transposed || g->K < 1 || g->K > 32 || g->ndim < 1 || g->ndim > WFS_MAX_DIM) return false;
    const int last = g->ndim - 1;
    if (last > 3) return false;
    G->ndim = g->ndim;
    G->K = g->K;
    G->kl = g->ksize[last];
    G->sl = g->stride[last];
    G->pl = g->padding[last];
    G->dl = g->dilation[last];
    G->out_last = g->out_shape[last];
    if (G->kl < 1 || G->kl > 8 || G->pl < 0) return false;
    if (!magic_ok(G->sl, g->spatial[last] + G->pl, &G->last_magic)) return false;
    long long cells = G->out_last;
    int kq = 1;
    for (int d = 0; d < 4; ++d) G->spatial[d] = d < g->ndim ? g->spatial[d] : 1;
    for (int d = 0; d < 3; ++d) {
        G->lks[d] = 1;
        G->ls[d] = 1;
        G->lp[d] = 0;
        G->ld[d] = 1;
        G->lout[d] = 1;
        G->lmult[d] = 0;
        G->lmagic[d] = 65536u;
        if (d >= last) continue;
        if (g->ksize[d] < 1 || g->ksize[d] > 4 || g->out_shape[d] < 1 || g->out_shape[d] > 255 || g->padding[d] < 0)
            return false;
        G->lks[d] = g->ksize[d];
        G->ls[d] = g->stride[d];
        G->lp[d] = g->padding[d];
        G->ld[d] = g->dilation[d];
        G->lout[d] = g->out_shape[d];
        if (!magic_ok(G->ls[d], g->spatial[d] + G->lp[d], &G->lmagic[d])) return false;
        cells *= g->out_shape[d];
        kq *= g->ksize[d];
    }
    if (cells < 1 || cells > EC_MAXCELLS || kq > 32 || kq * G->kl != g->K) return false;
    G->cells_e = (int)cells;
    G->Kq = kq;
    int mult = 1;
    for (int d = last - 1; d >= 0; --d) {
        G->lmult[d] = mult;
        mult *= g->out_shape[d];
    }
    for (int q = 0; q < 32; ++q) {
        G->qdig[q] = 0;
        if (q >= kq) continue;
        int rem = q;
        unsigned dg = 0;
        for (int d = last - 1; d >= 0; --d) {
            dg |= (unsigned)(rem % g->ksize[d]) << (8 * d);
            rem /= g->ksize[d];
        }
        G->qdig[q] = dg;
    }
    return true;
}

}  // namespace

extern "C" int wfs_event_rulebook_conv_ok(const wfs_geometry *g) {
    ECGeo G;
    return make_ecgeo(g, &G) ? 1 : 0;
}

// kl of the packed by-input table this geometry gets (0: dense [K, N])
extern "C" int wfs_event_rulebook_conv_packed_kl(const wfs_geometry *g) {
    ECGeo G;
    if (!make_ecgeo(g, &G)) return 0;
    return (G.kl >= 2 && G.kl <= G.sl && G.dl == 1) ? G.kl : 0;
}

extern "C" size_t wfs_event_rulebook_conv_state_bytes(int32_t batch_size) {
    return 64 + (size_t)(batch_size > 0 ? batch_size : 1) * sizeof(unsigned long long);
}

extern "C" int wfs_event_rulebook_conv(const wfs_geometry *g, const int32_t *indices, int64_t N, const int64_t *n_dev,
                                       const int32_t *events_in, int64_t M_cap, int32_t *out_indices, int64_t *m_dev,
                                       int32_t *events_out, int32_t *nbr_out, int32_t packed_kl, int32_t *nbr_in,
                                       int32_t *cell_row, int32_t *overflow_dev, int32_t *flags, void *state,
                                       void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ECGeo G;
    WFS_REQUIRE(make_ecgeo(g, &G), WFS_EINVAL, "wfs_event_rulebook_conv: geometry not covered (wfs_event_rulebook_conv_ok)");
    const int want_kl = wfs_event_rulebook_conv_packed_kl(g);
    WFS_REQUIRE(packed_kl == 0 || packed_kl == want_kl, WFS_EINVAL, "packed_kl %d: this geometry packs with %d", packed_kl,
                want_kl);
    WFS_REQUIRE(N >= 0 && (long long)g->K * N < (1ll << 31) && M_cap >= 1 && (long long)g->K * M_cap < (1ll << 31) &&
                    M_cap < (1ll << 28),
                WFS_EINVAL, "row counts out of range");
    WFS_REQUIRE(g->batch_size >= 1, WFS_EINVAL, "batch_size");
    WFS_REQUIRE(indices && n_dev && events_in && out_indices && m_dev && events_out && nbr_out && flags && state,
                WFS_EINVAL, "NULL device pointer");
    WFS_REQUIRE(((uintptr_t)state & 7) == 0, WFS_EINVAL, "state must be 8-byte aligned");
    WfsTimerScope timer(WFS_TIMER_RULEBOOK, stream);
    const int B = g->batch_size;
    const size_t lds = (size_t)G.cells_e * 6;
    unsigned *st = (unsigned *)state;
    unsigned long long *pub = (unsigned long long *)((char *)state + 64);
    const dim3 grid((unsigned)B), block(EC_THREADS);
    static bool attr_p = false, attr_d = false;
    if (packed_kl) {
        if (!attr_p && lds > 48 * 1024) {
            WFS_HIP_CHECK(hipFuncSetAttribute((const void *)k_ev_conv<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              EC_MAXCELLS * 6));
            attr_p = true;
        }
        k_ev_conv<true><<<grid, block, lds, stream>>>(G, indices, N, (const long long *)n_dev, events_in, B, M_cap,
                                                      out_indices, (long long *)m_dev, events_out, nbr_out, nbr_in, cell_row,
                                                      overflow_dev, flags, pub, st);
    } else {
        if (!attr_d && lds > 48 * 1024) {
            WFS_HIP_CHECK(hipFuncSetAttribute((const void *)k_ev_conv<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              EC_MAXCELLS * 6));
            attr_d = true;
        }
        k_ev_conv<false><<<grid, block, lds, stream>>>(G, indices, N, (const long long *)n_dev, events_in, B, M_cap,
                                                       out_indices, (long long *)m_dev, events_out, nbr_out, nbr_in,
                                                       cell_row, overflow_dev, flags, pub, st);
    }
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

extern "C" int wfs_unpack_table(const int32_t *packed, int32_t K, int32_t packed_kl, int64_t R, const int64_t *r_dev,
                                int32_t *dense, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(packed_kl >= 1 && packed_kl <= 8 && K >= 1 && K % packed_kl == 0, WFS_EINVAL, "bad K / packed_kl");
    if (R == 0) return WFS_OK;
    WFS_REQUIRE(packed && dense, WFS_EINVAL, "NULL device pointer");
    k_unpack_table<<<dim3((unsigned)wfs_cdiv(R, 256), (unsigned)(K / packed_kl)), dim3(256), 0, stream>>>(
        packed, K / packed_kl, packed_kl, R, (const long long *)r_dev, dense);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}
