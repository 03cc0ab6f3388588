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
//   k_ev_conv   regular / strided conv: candidates take tickets (row * K + offset) on an LDS grid of the event's output
//               sites with ds_min, a site's id is the rank of its first ticket (block scan over the event's rows), ids are
//               made global by a decoupled look-back over the events (one 64-bit word per event), then nbr_out, nbr_in,
//               out_indices and the cell -> row map are written once.  First-seen numbering as A.3 (bit-identical).
//
// Events are processed by workgroup (event mod gridDim) in increasing order, so a look-back only ever waits for events
// that have already been started: no deadlock for any batch size.
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
constexpr int ER_THREADS = 512;
constexpr int ER_MAXROWS = 2048;          // rows of one event the LDS tables cover (4 per thread)
constexpr int ER_RPT = ER_MAXROWS / ER_THREADS;
   // hash slots: load factor <= 1/4 (a miss ends at the first empty slot: the longest probe
                                          // sequence among a wave's lanes sets the pace, and it grows fast with the load)

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

// ------------------------------------------------------------------------------------------ SubM
// Site table of ONE event in LDS, direct addressing in two levels -- the detector's shape: a few active cells (leading
// dims: the PMT grid), each a dense run of samples (last dim: time):
//     cell_slot[cell]            -1 or the index of the cell's sample array      (L = prod(leading dims) entries)
//     pool[slot * T + t]         0 or 1 + local row                               (uint16; `nslot` arrays of T samples)
// A lookup is two dependent LDS reads, no probing, no key compare; the kl candidates of a row that differ only in the
// last dim's offset share the first.  (An open-addressing hash was measured first: at ~110 instructions per candidate --
// probe loops under divergence -- the largest event kept one CU busy for 30 us.)
// Capacity: L <= ER_MAXCELLS, active cells * T * 2 bytes <= ER_POOL bytes, rows <= 65534 per event; beyond that
// flags[0] is set and the caller takes rulebook.hip's build.  Duplicate coordinates are DETECTED (flags[1]; a row that
// does not read its own index back), not resolved: "the last row wins" (A.3) is then the caller's, i.e. rulebook.hip's.
// flags are SET, never cleared (the caller zeroes them once): [0] not grouped by event / capacity, [1] duplicates,
// [2] an index outside the spatial shape.
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
    if (!ev_structured(ev, B)) {
        if (threadIdx.x == 0) flags[0] = 1;
        return;
    }
    if (threadIdx.x < 32) sQt[threadIdx.x] = int2{qt.dcell[threadIdx.x], (int)qt.off[threadIdx.x]};
    const int cols = g.ndim + 1, last = g.ndim - 1;
    const int T = g.spatial[last];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // `split` workgroups per event: each builds the event's table and serves its share of the rows
    for (int item = blockIdx.x; item < B * split; item += gridDim.x) {
        const int e = item / split, part = item % split;
        const int o0 = ev[e];
        int o1 = ev[e + 1];
        o1 = o1 < Nv ? o1 : Nv;
        const int n = o1 - o0;
        if (n <= 0) continue;
        const int j_lo = (int)((long long)n * part / split), j_hi = (int)((long long)n * (part + 1) / split);
        if (n > 65534) {
            if (threadIdx.x == 0) flags[0] = 1;
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
                flags[2] = 1;
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
            if (threadIdx.x == 0) flags[0] = 1;
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
            if (cell >= 0 && pool[cell_slot[cell] * T + t] != (unsigned short)(j + 1)) flags[1] = 1;      // duplicate site
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
}

// ------------------------------------------------------------------------------------------ regular / strided conv
// Output sites of ONE event on a direct LDS grid (Vo = prod(out_shape) cells): tick[site] = smallest ticket
// (local row * K + offset) of the candidates that reach the site (ds_min), ids[site] = rank of that ticket among the
// event's first tickets = the site's first-seen number inside the event (A.3).  Two launches:
//   COUNT   passes 1-3 per event -> cnt[e] = number of output sites of event e
//   EMIT    passes 1-3 again (LDS only, a few us), base = sum of cnt[e' < e], then everything is written once: nbr_out,
//           nbr_in, out_indices, the outputs' event offsets, optionally the cell -> row map of dense() and the slot
//           records of the event-local conv's dX
// (a single launch would have to pass the bases between workgroups: a spin-wait on other workgroups' progress, a stamp
// that tells this launch's words from the last one's -- the recount is cheaper than either is safe.)
// Shapes: ndim <= 3, kernel 3 in every dim, Vo * 6 bytes of LDS + 2 K bytes per output row of an event (<= ER_CONV_LDS).  Inputs are taken to be distinct sites
// (a regular conv's input: checked by the SubM build of the same index set, or the output of another regular conv).
constexpr int ER_CONV_LDS = 156 * 1024;

// ONE: the last dim's stride is at least its kernel size (the PSD nets' k = 3, s = 4 layers), so a row reaches AT MOST ONE
// output cell along it -- through the offset (x + p) mod s, if that is below the kernel size: Q = prod(leading kernel
// dims) candidates per row instead of K.
template <int ND, bool ONE, bool EMIT>
__global__ void __launch_bounds__(ER_THREADS) k_ev_conv(EGeo g, int Vo, int split, const int *__restrict__ idx, long long N,
                                                        const long long *__restrict__ n_dev, const int *__restrict__ in_ev,
                                                        int B, int *__restrict__ cnt, int *__restrict__ nbr_out,
                                                        int *__restrict__ nbr_in, int *__restrict__ out_indices,
                                                        long long M_cap, int *__restrict__ out_ev, long long *info,
                                                        long long *m_dev, int *overflow, int *__restrict__ flags,
                                                        unsigned *__restrict__ cell_ticket, int *__restrict__ cell_row,
                                                        uint4 *__restrict__ slots_bwd, uint4 *__restrict__ slots_fwd,
                                                        int me_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char csm[];
    unsigned *tick = reinterpret_cast<unsigned *>(csm);
    unsigned short *ids = reinterpret_cast<unsigned short *>(csm + (size_t)Vo * 4);
    // EMIT: nin[k][id] = 1 + local input row that reaches output `id` of the event through offset k, or 0 (me_stride ids)
    unsigned short *nin = reinterpret_cast<unsigned short *>(csm + (((size_t)Vo * 6 + 15) & ~(size_t)15));
    __shared__ int sCount[ER_THREADS / 64 + 1];
    __shared__ int sBase;
    const int Nv = (int)valid_rows(N, n_dev);
    const bool structured = ev_structured(in_ev, B);
    if (!structured && threadIdx.x == 0) flags[0] = 1;
    constexpr int cols = ND + 1, LAST = ND - 1;
    constexpr int QA = ND >= 2 ? 3 : 1, QB = ND >= 3 ? 3 : 1;          // unrolled extents of the leading kernel dims
    const int K = g.K;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int ksl = g.ksize[LAST], sl = g.stride[LAST], outl = g.out_shape[LAST];
    const int ka = ND >= 2 ? 3 : 1, kb = ND >= 3 ? 3 : 1;          // kernel 3 in every dim (wfs_event_rulebook_conv_ok)
    const int sh_l = (sl & (sl - 1)) == 0 ? __builtin_ctz(sl) : -1;

    // output coordinate reached by input coordinate xd through offset o of LEADING dim d, or -1
    auto lead_coord = [&](int d, int xd, int o) -> int {
        const int tt = xd + g.padding[d] - o * g.dilation[d];
        if (o >= g.ksize[d] || tt < 0) return -1;
        const int st = g.stride[d];
        const int q = st == 1 ? tt : (int)((unsigned)tt / (unsigned)st);
        return (q * st == tt && q < g.out_shape[d]) ? q : -1;
    };
    struct Cand {
        int lead[QA * QB];        // leading part of the site (row-major over the leading out dims) per leading offset pair, or -1
        int ot[ONE ? 1 : 3];      // output coordinate along the last dim per candidate offset, or -1
        int cv;                   // ONE: the offset along the last dim (or -1)
    };
    auto candidates = [&](const int *x, Cand &c) {
#pragma unroll
        for (int a = 0; a < QA; ++a)
#pragma unroll
            for (int b = 0; b < QB; ++b) {
                int site = 0;
                bool ok = true;
                if (ND >= 2) {
                    const int oa = lead_coord(0, x[0], a);
                    ok = oa >= 0;
                    site = oa;
                }
                if (ND >= 3) {
                    const int ob = lead_coord(1, x[1], b);
                    ok = ok && ob >= 0;
                    site = site * g.out_shape[1] + ob;
                }
                c.lead[a * QB + b] = ok ? site : -1;
            }
        const int xl = x[LAST] + g.padding[LAST];
        if constexpr (ONE) {
            // dilation 1 here (stride > 1 excludes dilation > 1, A.1): offset o reaches (xl - o) / s when s divides it
            const int r0 = sh_l >= 0 ? (xl & (sl - 1)) : (int)((unsigned)xl % (unsigned)sl);
            const int tt = xl - r0;
            const int q = sh_l >= 0 ? (tt >> sh_l) : (int)((unsigned)tt / (unsigned)sl);
            const bool ok = xl >= 0 && r0 < ksl && tt >= 0 && q < outl;
            c.cv = ok ? r0 : -1;
            c.ot[0] = q;
        } else {
            c.cv = -1;
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const int tt = xl - o * g.dilation[LAST];
                const int q = sl == 1 ? tt : (int)((unsigned)(tt < 0 ? 0 : tt) / (unsigned)sl);
                c.ot[o] = (o < ksl && tt >= 0 && q * sl == tt && q < outl) ? q : -1;
            }
        }
    };
    // f(k, site) for every candidate that reaches an output site, in increasing k
    auto for_sites = [&](const Cand &c, auto f) {
#pragma unroll
        for (int a = 0; a < QA; ++a)
#pragma unroll
            for (int b = 0; b < QB; ++b) {
                if (a >= ka || b >= kb) continue;
                const int ls = c.lead[a * QB + b];
                const int kq = (a * kb + b) * ksl;
                if constexpr (ONE) {
                    if (ls >= 0 && c.cv >= 0) f(kq + c.cv, ls * outl + c.ot[0]);
                } else {
#pragma unroll
                    for (int o = 0; o < 3; ++o)
                        if (ls >= 0 && c.ot[o] >= 0) f(kq + o, ls * outl + c.ot[o]);
                }
            }
    };
    auto load_x = [&](int row, int *x) -> bool {
        const int *r = idx + (long long)row * cols;
        bool ok = true;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            x[d] = r[1 + d];
            ok = ok && x[d] >= 0 && x[d] < g.spatial[d];
        }
        return ok;
    };

    for (int item = blockIdx.x; item < B * split; item += gridDim.x) {
        const int e = item / split, part = item - e * split;
        int i0 = 0, n = 0;
        if (structured) {
            i0 = in_ev[e];
            int i1 = in_ev[e + 1];
            i1 = i1 < Nv ? i1 : Nv;
            n = i1 - i0;
            n = n > 0 ? n : 0;
        }
        if ((long long)n * K >= (1ll << 32)) {
            if (threadIdx.x == 0) flags[0] = 1;
            n = 0;
        }
        __syncthreads();                                      // the previous event's grid is no longer read
        for (int s_ = threadIdx.x; s_ < Vo; s_ += ER_THREADS) tick[s_] = 0xFFFFFFFFu;
        __syncthreads();
        // pass 2: tickets
#pragma unroll 1
        for (int j = threadIdx.x; j < n; j += ER_THREADS) {
            int x[3];
            if (!load_x(i0 + j, x)) {
                flags[2] = 1;
                continue;
            }
            Cand c;
            candidates(x, c);
            for_sites(c, [&](int k, int site) { atomicMin(&tick[site], (unsigned)(j * K + k)); });
        }
        __syncthreads();
        // pass 3: first tickets -> ids (512 consecutive rows per round: a block scan gives their bases in row order)
        int carry = 0;
        for (int j0 = 0; j0 < n; j0 += ER_THREADS) {
            const int j = j0 + threadIdx.x;
            int x[3];
            const bool live = j < n && load_x(i0 + j, x);
            Cand c;
            unsigned mask = 0;
            if (live) {
                candidates(x, c);
                for_sites(c, [&](int k, int site) {
                    if (tick[site] == (unsigned)(j * K + k)) mask |= 1u << k;
                });
            }
            const int cn = __popc(mask);
            int incl = cn;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(incl, d, 64);
                if (lane >= d) incl += v;
            }
            if (lane == 63) sCount[wid] = incl;
            __syncthreads();
            int base = carry, tot = 0;
#pragma unroll
            for (int w = 0; w < ER_THREADS / 64; ++w) {
                const int cw = sCount[w];
                if (w < wid) base += cw;
                tot += cw;
            }
            if (EMIT && mask) {
                const int rowbase = base + incl - cn;
                for_sites(c, [&](int k, int site) {
                    if ((mask >> k) & 1u) ids[site] = (unsigned short)(rowbase + __popc(mask & ((1u << k) - 1u)));
                });
            }
            carry += tot;
            __syncthreads();
        }
        const int Me = carry;
        if (Me > 65535 && threadIdx.x == 0) flags[0] = 1;          // ids are 16 bits
        if constexpr (!EMIT) {
            if (threadIdx.x == 0) cnt[e] = Me;
            continue;
        } else {
            // base of this event's outputs = sum of the counts of the events in front (fixed order: thread-strided, a wave
            // reduction, then the waves in order)
            int partial = 0;
            for (int q = threadIdx.x; q < e; q += ER_THREADS) partial += cnt[q];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) partial += __shfl_xor(partial, d, 64);
            if (lane == 0) sCount[wid] = partial;
            __syncthreads();
            if (threadIdx.x == 0) {
                int b_ = 0;
                for (int w = 0; w < ER_THREADS / 64; ++w) b_ += sCount[w];
                sBase = b_;
            }
            __syncthreads();
            const long long base = sBase;
            if (threadIdx.x == 0 && part == 0) {
                out_ev[e] = (int)(base < M_cap ? base : M_cap);
                if (e == B - 1) {
                    const long long M = base + Me;
                    out_ev[B] = (int)(M < M_cap ? M : M_cap);
                    if (info) info[0] = M;
                    if (m_dev) *m_dev = M < M_cap ? M : M_cap;
                    if (overflow) *overflow = M > M_cap ? 1 : 0;
                }
            }
            if (item == 0 && threadIdx.x < WFS_EVENT_FLAG_WORDS) out_ev[B + 1 + threadIdx.x] = structured ? 0 : 1;
            // this workgroup's share of the event's rows, output rows and cells
            const int j_lo = (int)((long long)n * part / split), j_hi = (int)((long long)n * (part + 1) / split);
            const int m_lo = (int)((long long)Me * part / split), m_hi = (int)((long long)Me * (part + 1) / split);
            const int v_lo = (int)((long long)Vo * part / split), v_hi = (int)((long long)Vo * (part + 1) / split);
            if (Me > me_stride) {                      // more outputs than the LDS image of nbr_in holds
                if (threadIdx.x == 0) flags[0] = 1;
                continue;
            }
            // pass 4a: clear the LDS image of this event's nbr_in columns
            {
                unsigned *n32 = reinterpret_cast<unsigned *>(nin);
                const int nd = (K * me_stride) >> 1;
                for (int q = threadIdx.x; q < nd; q += ER_THREADS) n32[q] = 0u;
            }
            __syncthreads();
            // pass 4b: from the input side.  Every workgroup of the event fills the WHOLE LDS image (LDS scatter is cheap);
            // the global tables of a row are written by the workgroup whose share the row is.
#pragma unroll 1
            for (int j = threadIdx.x; j < ((ER_KNOCK & 64) ? 0 : n); j += ER_THREADS) {
                const bool mine = j >= j_lo && j < j_hi;
                int x[3];
                const bool okx = load_x(i0 + j, x);
                const int row = i0 + j;
                Cand c;
                candidates(x, c);
                unsigned rec[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) rec[q] = 0u;
#pragma unroll
                for (int a = 0; a < QA; ++a)
#pragma unroll
                    for (int b = 0; b < QB; ++b) {
                        const int ls = okx ? c.lead[a * QB + b] : -1;
#pragma unroll
                        for (int o = 0; o < 3; ++o) {
                            constexpr int dummy = 0;
                            (void)dummy;
                            const int k = (a * QB + b) * 3 + o;              // kernel 3 in every dim: k is a constant here
                            const int otc = ONE ? (o == c.cv ? c.ot[0] : -1) : c.ot[ONE ? 0 : o];
                            int gid = -1;
                            unsigned lid1 = 0;
                            if (ls >= 0 && otc >= 0) {
                                const int site = ls * outl + otc;
                                const int lid = ids[site];
                                nin[k * me_stride + lid] = (unsigned short)(j + 1);
                                const long long gg = base + lid;
                                if (gg < M_cap) {
                                    gid = (int)gg;
                                    lid1 = (unsigned)lid + 1u;
                                    if (mine && tick[site] == (unsigned)(j * K + k)) {      // first ticket: the row introduces the site
                                        int oi[4] = {e, 0, 0, 0};
                                        int rem = site;
#pragma unroll
                                        for (int d = ND - 1; d >= 0; --d) {
                                            oi[1 + d] = rem % g.out_shape[d];
                                            rem /= g.out_shape[d];
                                        }
                                        int *dst = out_indices + gg * cols;
#pragma unroll
                                        for (int d = 0; d < cols; ++d) dst[d] = oi[d];
                                    }
                                }
                            }
                            if (mine) {
                                int *col = nbr_out + (long long)k * N;
                                col[(unsigned)row] = gid;
                            }
                            rec[k >> 1] |= (k & 1) ? (lid1 << 16) : lid1;
                        }
                    }
                if (slots_bwd && mine) {
                    uint4 *dst = slots_bwd + (long long)row * 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) dst[q] = uint4{rec[4 * q], rec[4 * q + 1], rec[4 * q + 2], rec[4 * q + 3]};
                }
            }
            __syncthreads();
            // pass 4c: from the output side: nbr_in columns (coalesced over the ids) and the forward slot records
            if (nbr_in && !(ER_KNOCK & 128)) {
                for (int k = 0; k < K; ++k) {
                    int *col = nbr_in + (long long)k * M_cap;
                    for (int id = m_lo + threadIdx.x; id < m_hi; id += ER_THREADS) {
                        const long long gg = base + id;
                        const unsigned v = nin[k * me_stride + id];
                        if (gg < M_cap) col[gg] = v ? i0 + (int)v - 1 : -1;
                    }
                }
            }
            if (slots_fwd) {
                for (int id = m_lo + threadIdx.x; id < m_hi; id += ER_THREADS) {
                    const long long gg = base + id;
                    if (gg >= M_cap) continue;
                    unsigned rec[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) rec[q] = 0u;
#pragma unroll
                    for (int k = 0; k < 27; ++k) {
                        if (k >= K) break;
                        const unsigned v = nin[k * me_stride + id];
                        rec[k >> 1] |= (k & 1) ? (v << 16) : v;
                    }
                    uint4 *dst = slots_fwd + gg * 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) dst[q] = uint4{rec[4 * q], rec[4 * q + 1], rec[4 * q + 2], rec[4 * q + 3]};
                }
            }
            // pass 4c: cell -> row map of this event (dense() of the outputs)
            if (cell_ticket) {
                for (int s_ = v_lo + threadIdx.x; s_ < v_hi; s_ += ER_THREADS) {
                    const unsigned tk = tick[s_];
                    const long long gg = base + ids[s_];
                    const bool act = tk != 0xFFFFFFFFu && gg < M_cap;
                    cell_ticket[(long long)e * Vo + s_] = act ? tk : 0xFFFFFFFFu;
                    cell_row[(long long)e * Vo + s_] = act ? (int)gg : -1;
                }
            }
        }
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
    // sample arrays for 48 active cells per event (a PSD event has <= ~20), within 8 .. 64 KiB; with a small table four
    // workgroups fit a compute unit and share an event, else two
    const int T = g->spatial[g->ndim - 1];
    long long pool_ll = 48ll * T * 2;
    pool_ll = pool_ll < 8192 ? 8192 : (pool_ll > ER_POOL ? ER_POOL : pool_ll);
    const int pool_bytes = (int)((pool_ll + 15) / 16 * 16);
    const int split = 2;          // measured at the PSD batch: 1 -> 15.6 us, 2 -> 12.7 us, 4 -> 18.9 us
    const long long items = (long long)B * split;
    const int nblk = (int)(items < 2048 ? items : 2048);
    int Q = 1, L = 1;
    const EQTab qt = make_qtab_subm(g, &Q, &L);
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

extern "C" size_t wfs_event_rulebook_conv_workspace_bytes(int32_t batch_size) {
    return (size_t)(batch_size > 0 ? batch_size : 1) * sizeof(int32_t);
}

extern "C" int wfs_event_rulebook_conv_ok(const wfs_geometry *g) {
    if (!g || g->subm || g->transposed || g->K < 1 || g->K > 27 || g->ndim < 1 || g->ndim > 3) return 0;
    long long vo = 1;
    for (int d = 0; d < g->ndim; ++d) {
        if (g->ksize[d] != 3 || g->stride[d] < 1) return 0;
        vo *= g->out_shape[d];
    }
    // LDS: the ticket grid (6 bytes per output cell) + the image of one event's nbr_in (2 K bytes per output row): room for
    // at least 1024 output rows per event
    return vo * 6 + 16 + (long long)g->K * 2 * 1024 <= ER_CONV_LDS;
}

extern "C" int wfs_event_rulebook_conv(const wfs_geometry *g, const int32_t *indices, int64_t N, const int64_t *n_dev,
                                       const int32_t *in_events, int32_t *nbr_out, int32_t *nbr_in, int32_t *out_indices,
                                       int64_t M_cap, int32_t *out_events, int64_t *info, int64_t *m_dev,
                                       int32_t *overflow_dev, int32_t *flags, uint32_t *cell_ticket, int32_t *cell_row,
                                       void *slots_bwd, void *slots_fwd, void *workspace, size_t workspace_bytes,
                                       void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(g && wfs_event_rulebook_conv_ok(g), WFS_EINVAL,
                "wfs_event_rulebook_conv: regular conv, ndim <= 3, kernel <= 3 per dim, out volume * 6 B of LDS");
    WFS_REQUIRE(N >= 0 && (long long)g->K * N < (1ll << 31) && M_cap >= 0 && (long long)g->K * M_cap < (1ll << 31),
                WFS_EINVAL, "N / M_cap out of range");
    const int B = g->batch_size;
    WFS_REQUIRE(workspace && workspace_bytes >= wfs_event_rulebook_conv_workspace_bytes(B), WFS_EWORKSPACE,
                "workspace too small");
    WFS_REQUIRE(in_events && out_events && flags && nbr_out && (indices || N == 0), WFS_EINVAL, "NULL device pointer");
    WFS_REQUIRE((cell_ticket == nullptr) == (cell_row == nullptr), WFS_EINVAL, "cell_ticket and cell_row come together");
    WfsTimerScope timer(WFS_TIMER_RULEBOOK, stream);
    long long vo = 1;
    for (int d = 0; d < g->ndim; ++d) vo *= g->out_shape[d];
    const int Vo = (int)vo;
    const size_t grid_bytes = ((size_t)Vo * 6 + 15) & ~(size_t)15;
    const int me_stride = (int)(((size_t)ER_CONV_LDS - grid_bytes) / (2 * (size_t)g->K)) & ~7;
    const EGeo G = make_egeo(g);
    int *cnt = (int *)workspace;
    const int last = g->ndim - 1;
    const bool one = g->stride[last] >= g->ksize[last] && g->dilation[last] == 1;
    static const int split = [] { const char *e_ = getenv("WFS_EVRB_SPLIT"); return e_ ? atoi(e_) : 1; }();
    const dim3 block(ER_THREADS);
    static bool attr_done[3][2][2] = {};
#define WFS_EVC(ND, ONE_, EM)                                                                                            \
    do {                                                                                                                 \
        auto kern = k_ev_conv<ND, ONE_, EM>;                                                                             \
        if (!attr_done[ND - 1][ONE_ ? 1 : 0][EM ? 1 : 0]) {                                                              \
            WFS_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,            \
                                              ER_CONV_LDS));                                                             \
            attr_done[ND - 1][ONE_ ? 1 : 0][EM ? 1 : 0] = true;                                                          \
        }                                                                                                                \
        const int sp = EM ? split : 1;                                                                                   \
        const long long items = (long long)B * sp;                                                                       \
        const dim3 grid((unsigned)(items < 2048 ? items : 2048));                                                        \
        const size_t lds = EM ? grid_bytes + (size_t)g->K * me_stride * 2 : grid_bytes;                                  \
        kern<<<grid, block, lds, stream>>>(G, Vo, sp, indices, N, (const long long *)n_dev, in_events, B, cnt, nbr_out,  \
                                           nbr_in, out_indices, M_cap, out_events, (long long *)info,                    \
                                           (long long *)m_dev, overflow_dev, flags, cell_ticket, cell_row,               \
                                           (uint4 *)slots_bwd, (uint4 *)slots_fwd, me_stride);                           \
        WFS_LAUNCH_CHECK();                                                                                              \
    } while (0)
#define WFS_EVC2(ND)                                                                                                     \
    do {                                                                                                                 \
        if (one) {                                                                                                       \
            if (!(ER_KNOCK & 32)) WFS_EVC(ND, true, false);                                                              \
            if (!(ER_KNOCK & 16)) WFS_EVC(ND, true, true);                                                               \
        } else {                                                                                                         \
            if (!(ER_KNOCK & 32)) WFS_EVC(ND, false, false);                                                             \
            if (!(ER_KNOCK & 16)) WFS_EVC(ND, false, true);                                                              \
        }                                                                                                                \
    } while (0)
    if (g->ndim == 1)
        WFS_EVC2(1);
    else if (g->ndim == 2)
        WFS_EVC2(2);
    else
        WFS_EVC2(3);
#undef WFS_EVC2
#undef WFS_EVC
    return WFS_OK;
}
