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
//   4  every candidate reads its site's id from LDS and the tables are written ONCE, coalesced (the by-output table
//      through an LDS image, the coordinates one thread per site): out_indices, the by-input
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
// minimum waves per SIMD the build is compiled for = its register cap.  7 -> 72 registers: a 512-thread workgroup then
// holds 2 x 72 of a SIMD's 512 registers, and what runs BESIDE the builds in a training step still fits the same SIMD --
// the register-resident BatchNorm kernels (200 registers) and the 12-wave conv blocks (3 x 120).  Same-box A/B of the
// captured step (tools/exp/ab_multi.sh): 1024 threads uncapped 0.4825 / 0.4884 / 0.4983 ms, 512 threads capped at 72
// registers 0.4727 / 0.4810, at 64 registers (36 B of scratch) 0.4766 / 0.4774
#ifndef EC_MIN_WAVES
#define EC_MIN_WAVES 7
#endif
// threads per workgroup: 1024 (one row per thread for all but the largest events)
constexpr int EC_MAXCELLS = 16384;            // output cells of one event: 4 B ticket / image + 2 B id in LDS
constexpr int EC_FLAG_WORDS = WFS_EVENT_FLAG_WORDS;
constexpr unsigned EC_SPIN_LIMIT = 1u << 22;  // polls of a predecessor's count before the launch gives up (flagged)
// every count is published EC_REPLICAS times, EC_REPLICA_STRIDE(B) bytes apart, and event e reads replica e % EC_REPLICAS:
// all events behind a slow one poll ITS word, and polls of one word queue at one memory channel
constexpr int EC_REPLICAS = 8;
inline __host__ __device__ size_t ec_replica_stride(int B) { return ((size_t)B * 8 + 4095) / 4096 * 4096 + 256; }

struct ECGeo {
    int ndim, K, Kq, kl, sl, pl, dl, out_last, cells_e;
    int spatial[4];
    // leading dims (all but the last; unused ones: ksize 1, out 1, mult 0)
    int lks[3], ls[3], lp[3], ld[3], lout[3], lmult[3];
    unsigned lmagic[3], last_magic;          // floor(t / s) == (t * magic) >> 16 for the t that occur (host-verified)
    unsigned dmagic[4];                      // floor(c / out_shape[d]) == umulhi(c, dmagic[d]) for c < 2^16 (cell decoding)
    int oshape[4];
    // candidate slots: packed form: slot = leading-offset index q (the last dim's offset follows from the row);
    // dense form: slot = kernel offset k.  Digits of the offsets, 8 bits per dim (leading dims 0..2, last dim in 24..31)
    unsigned dig[32];
};

__device__ __forceinline__ long long valid_rows(long long R, const long long *r_dev) {
    long long v = r_dev ? *r_dev : R;
    return v < R ? v : R;
}

// The candidates of one input row, in registers: slot t (compile-time index) holds the event-local output cell the
// row reaches through that slot's kernel offset, or -1.  PACKED: o_star = the one offset along the last dim that can
// divide (kernel <= stride there), so the kernel offset of slot q is q * kl + o_star.
template <int NQ>
struct Cand {
    int cell[NQ];
    int o_star;
    bool ok;                 // coordinates inside the spatial shape
};

template <bool PACKED, int NQ>
__device__ __forceinline__ Cand<NQ> digest_row(const ECGeo &g, const int *__restrict__ row) {
    Cand<NQ> c;
    bool ok = true;
    const int last = g.ndim - 1;
    unsigned vm[3], pk[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        vm[d] = 1u;
        pk[d] = 0u;
        if (d < last) {
            const int x = row[1 + d];
            ok = ok && x >= 0 && x < g.spatial[d];
            unsigned m = 0u, p = 0u;
            for (int o = 0; o < g.lks[d]; ++o) {
                const int t = x + g.lp[d] - o * g.ld[d];
                const int oc = (int)(((unsigned)(t < 0 ? 0 : t) * g.lmagic[d]) >> 16);
                const bool v = t >= 0 && oc * g.ls[d] == t && oc < g.lout[d];
                m |= v ? (1u << o) : 0u;
                p |= (unsigned)(v ? oc : 0) << (8 * o);
            }
            vm[d] = m;
            pk[d] = p;
        }
    }
    const int xl = row[1 + last];
    ok = ok && xl >= 0 && xl < g.spatial[last];
    const int tl = xl + g.pl;
    c.ok = ok;
    c.o_star = 0;
    int oc_star = 0;
    bool l_ok = true;
    if (PACKED) {
        // kernel <= stride along the last dim (dilation 1): only offset (x + p) mod s can divide
        oc_star = (int)(((unsigned)(tl < 0 ? 0 : tl) * g.last_magic) >> 16);
        c.o_star = tl - oc_star * g.sl;
        l_ok = c.o_star < g.kl && oc_star < g.out_last;
    }
    const int nslots = PACKED ? g.Kq : g.K;
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        const unsigned dg = g.dig[t];
        const unsigned o0 = dg & 255u, o1 = (dg >> 8) & 255u, o2 = (dg >> 16) & 255u;
        bool v = ok && t < nslots && (((vm[0] >> o0) & (vm[1] >> o1) & (vm[2] >> o2) & 1u) != 0u);
        const int lead = (int)((pk[0] >> (8 * o0)) & 255u) * g.lmult[0] + (int)((pk[1] >> (8 * o1)) & 255u) * g.lmult[1] +
                         (int)((pk[2] >> (8 * o2)) & 255u) * g.lmult[2];
        int oc = oc_star;
        if (PACKED) {
            v = v && l_ok;
        } else {
            const int tt = tl - (int)(dg >> 24) * g.dl;
            oc = (int)(((unsigned)(tt < 0 ? 0 : tt) * g.last_magic) >> 16);
            v = v && tt >= 0 && oc * g.sl == tt && oc < g.out_last;
        }
        c.cell[t] = v ? lead * g.out_last + oc : -1;
    }
    return c;
}

// The same enumeration without the register cache, for geometries with more slots than fit one (NQ > 9): calls
// f(t, cell, k) for every slot of the row that reaches an output cell, in increasing t
template <bool PACKED, typename F>
__device__ __forceinline__ void stream_row(const ECGeo &g, const int *__restrict__ row, bool *ok_out, F f) {
    bool ok = true;
    const int last = g.ndim - 1;
    unsigned vm[3], pk[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        vm[d] = 1u;
        pk[d] = 0u;
        if (d < last) {
            const int x = row[1 + d];
            ok = ok && x >= 0 && x < g.spatial[d];
            unsigned m = 0u, p = 0u;
            for (int o = 0; o < g.lks[d]; ++o) {
                const int t = x + g.lp[d] - o * g.ld[d];
                const int oc = (int)(((unsigned)(t < 0 ? 0 : t) * g.lmagic[d]) >> 16);
                const bool v = t >= 0 && oc * g.ls[d] == t && oc < g.lout[d];
                m |= v ? (1u << o) : 0u;
                p |= (unsigned)(v ? oc : 0) << (8 * o);
            }
            vm[d] = m;
            pk[d] = p;
        }
    }
    const int xl = row[1 + last];
    ok = ok && xl >= 0 && xl < g.spatial[last];
    *ok_out = ok;
    if (!ok) return;
    const int tl = xl + g.pl;
    int o_star = 0, oc_star = 0;
    bool l_ok = true;
    if (PACKED) {
        oc_star = (int)(((unsigned)tl * g.last_magic) >> 16);
        o_star = tl - oc_star * g.sl;
        l_ok = o_star < g.kl && oc_star < g.out_last;
        if (!l_ok) return;
    }
    const int nslots = PACKED ? g.Kq : g.K;
#pragma unroll 1
    for (int t = 0; t < nslots; ++t) {
        const unsigned dg = g.dig[t];
        const unsigned o0 = dg & 255u, o1 = (dg >> 8) & 255u, o2 = (dg >> 16) & 255u;
        if (!((vm[0] >> o0) & (vm[1] >> o1) & (vm[2] >> o2) & 1u)) continue;
        const int lead = (int)((pk[0] >> (8 * o0)) & 255u) * g.lmult[0] + (int)((pk[1] >> (8 * o1)) & 255u) * g.lmult[1] +
                         (int)((pk[2] >> (8 * o2)) & 255u) * g.lmult[2];
        if (PACKED) {
            f(t, lead * g.out_last + oc_star, t * g.kl + o_star);
        } else {
            const int tt = tl - (int)(dg >> 24) * g.dl;
            if (tt < 0) continue;
            const int oc = (int)(((unsigned)tt * g.last_magic) >> 16);
            if (oc * g.sl == tt && oc < g.out_last) f(t, lead * g.out_last + oc, t);
        }
    }
}

// block-wide exclusive scan of one int per thread (threads in order); *total = the block's sum
// Barrier for data exchanged through LDS only: waits for this wave's LDS traffic, NOT for its outstanding global loads
// / stores (__syncthreads() would: the look-back's loads and the table stores are meant to stay in flight across it)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int EC_WAVES>
__device__ __forceinline__ int ec_block_scan(int v, int *sWave, int *total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int n = __shfl_up(inc, d, 64);
        if (lane >= d) inc += n;
    }
    lds_barrier();                        // the previous use of sWave is over
    if (lane == 63) sWave[wid] = inc;
    lds_barrier();
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

template <bool PACKED, int NQ, int EC_THREADS>
__global__ void __launch_bounds__(EC_THREADS) __attribute__((amdgpu_waves_per_eu(EC_THREADS <= 512 ? EC_MIN_WAVES : 4))) k_ev_conv(ECGeo g, int img_bytes, const int *__restrict__ idx, long long N,
                                                        const long long *__restrict__ n_dev,
                                                        const int *__restrict__ ev_in, int B, long long M_cap,
                                                        int *__restrict__ out_idx, long long *__restrict__ m_dev,
                                                        int *__restrict__ ev_out, int *__restrict__ nbr_out,
                                                        int *__restrict__ nbr_in, int *__restrict__ cell_row,
                                                        int *__restrict__ overflow, int *__restrict__ flags,
                                                        unsigned long long *__restrict__ pub,
                                                        unsigned *__restrict__ state) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ec_lds[];
    // LDS: [ tickets (4 B per cell), later the by-output image | cell -> id (2 B) ].  The image takes the tickets' place
    // once the ids exist (img_bytes >= 4 * cells_e): a workgroup of the PSD geometry holds 41 KB -- three to a CU when a
    // batch has more events than the chip has CUs, and the 96-KB blocks of the layers' conv kernels, which run beside the
    // builds in a training step, still find room on the same CU
    unsigned *ticket = reinterpret_cast<unsigned *>(ec_lds);                                        // [cells_e]
    unsigned short *img = reinterpret_cast<unsigned short *>(ec_lds);                               // [group][M_e (even)]
    unsigned short *cellid = reinterpret_cast<unsigned short *>(ec_lds + (size_t)img_bytes);       // [cells_e] cell -> id in the event (0xFFFF: not an output site)
    constexpr int EC_WAVES = EC_THREADS / 64;
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
    int n = structured && r1 > r0 ? r1 - r0 : 0;
    int f_fail = 0;
    if (n > 65534) {                                    // local rows travel as uint16 through the image
        n = 0;
        f_fail = 1;
    }
    int f_range = 0;
    const int nslots = PACKED ? g.Kq : g.K;

    for (int c = tid; c < g.cells_e; c += EC_THREADS) {
        ticket[c] = 0xFFFFFFFFu;
        cellid[c] = 0xFFFFu;
    }
    // CACHE (<= 9 candidate slots): this thread's first row stays digested in registers through all phases (events
    // beyond EC_THREADS rows re-digest the further ones); wider geometries stream their candidates in every phase
    constexpr bool CACHE = NQ <= 9;
    constexpr int NC = CACHE ? NQ : 1;
    Cand<NC> mine;
#pragma unroll
    for (int t = 0; t < NC; ++t) mine.cell[t] = -1;
    mine.o_star = 0;
    mine.ok = true;
    if (CACHE && tid < n) mine = digest_row<PACKED, NC>(g, idx + (long long)(r0 + tid) * cols);
    // visit(jl, f): f(t, cell, k) for every candidate of local row jl that reaches an output cell, in increasing t
    auto visit = [&](int jl, auto f) {
        if constexpr (CACHE) {
            Cand<NC> c = mine;
            if (jl != tid) c = digest_row<PACKED, NC>(g, idx + (long long)(r0 + jl) * cols);
            if (!c.ok) f_range = 1;
#pragma unroll
            for (int t = 0; t < NC; ++t)
                if (c.cell[t] >= 0) f(t, c.cell[t], PACKED ? t * g.kl + c.o_star : t);
        } else {
            bool ok;
            stream_row<PACKED>(g, idx + (long long)(r0 + jl) * cols, &ok, f);
            if (!ok) f_range = 1;
        }
    };
    lds_barrier();
    // ---- 1: tickets
    for (int jl = tid; jl < n && !(EC_KNOCK & 8); jl += EC_THREADS)
        visit(jl, [&](int, int cell, int k) { atomicMin(&ticket[cell], (unsigned)jl * 32u + (unsigned)k); });
    lds_barrier();
    // ---- 2: first flags -> ids within the event (cellid)
    int carry = 0;
    for (int j0 = 0; j0 < n && !(EC_KNOCK & 8); j0 += EC_THREADS) {
        const int jl = j0 + tid;
        unsigned mask = 0u;
        if (jl < n)
            visit(jl, [&](int t, int cell, int k) {
                mask |= ticket[cell] == (unsigned)jl * 32u + (unsigned)k ? (1u << t) : 0u;
            });
        int tot;
        const int rowbase = carry + ec_block_scan<EC_WAVES>(__popc(mask), sWave, &tot);
        if (mask != 0u)
            visit(jl, [&](int t, int cell, int) {
                if ((mask >> t) & 1u) {
                    const int il = rowbase + __popc(mask & ((1u << t) - 1u));
                    cellid[cell] = (unsigned short)il;
                }
            });
        carry += tot;
    }
    const int M_e = carry;
    // ---- 3: publish the count; the counts in front are asked for now and looked at after the work that does not need
    // them (the first pass of the by-output image, the ids of this thread's candidates)
    const size_t rstride = ec_replica_stride(B) / 8;
    if (tid < EC_REPLICAS)
        __hip_atomic_store(&pub[tid * rstride + e], ((unsigned long long)tag << 32) | (unsigned long long)(unsigned)M_e,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long *mypub = pub + (size_t)(e % EC_REPLICAS) * rstride;
    unsigned long long w_first = 0ull;
    if (tid < e && !(EC_KNOCK & 1)) w_first = __hip_atomic_load(&mypub[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lds_barrier();                                    // cellid is complete
    // by-output table through an LDS image [offsets of the pass][M_e]: every row of nbr_in is written once, coalesced,
    // values and "none" alike (no -1 fill behind it, no scattered 4-byte stores)
    if (EC_KNOCK & (2 | 4)) nbr_in = nullptr;
    const int mstride = (M_e + 1) & ~1;
    int gs = M_e > 0 ? img_bytes / (2 * mstride) : g.K;  // offsets per pass (>= 1: the image holds 2 * cells_e bytes at least)
    gs = gs < 1 ? 1 : (gs > g.K ? g.K : gs);
    auto image_pass = [&](int k0, int k1) {             // fills the image of offsets [k0, k1); ends with a barrier
        unsigned *img32 = reinterpret_cast<unsigned *>(img);
        const int nd = ((k1 - k0) * mstride) >> 1;
        for (int i = tid; i < nd; i += EC_THREADS) img32[i] = 0xFFFFFFFFu;
        lds_barrier();
        for (int jl = tid; jl < n; jl += EC_THREADS)
            visit(jl, [&](int, int cell, int k) {
                if (k >= k0 && k < k1) img[(k - k0) * mstride + (int)cellid[cell]] = (unsigned short)jl;
            });
        lds_barrier();
    };
    if (nbr_in && M_e > 0) image_pass(0, gs < g.K ? gs : g.K);
    // ids (within the event) of this thread's cached row
    int il_mine[NC];
#pragma unroll
    for (int t = 0; t < NC; ++t) il_mine[t] = (CACHE && tid < n && mine.cell[t] >= 0) ? (int)cellid[mine.cell[t]] : -1;
    long long part = 0;
    int timed_out = 0;
    for (int p = tid; p < e && !(EC_KNOCK & 1); p += EC_THREADS) {
        unsigned long long w = p == tid ? w_first : __hip_atomic_load(&mypub[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((unsigned)(w >> 32) != tag && spins < EC_SPIN_LIMIT) {
            __builtin_amdgcn_s_sleep(1);
            w = __hip_atomic_load(&mypub[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ++spins;
        }
        if ((unsigned)(w >> 32) != tag) timed_out = 1;
        part += (long long)(unsigned)(w & 0xFFFFFFFFull);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    if (lane == 0) sSum[wid] = part;
    lds_barrier();
    long long base = 0;
#pragma unroll
    for (int w = 0; w < EC_WAVES; ++w) base += sSum[w];
    const long long m_total = base + M_e;               // meaningful in the last event's workgroup
    const long long m_room = base < M_cap ? M_cap - base : 0;        // ids of this event below m_room exist
    const int mw = (long long)M_e < m_room ? M_e : (int)m_room;     // rows of this event that exist
    // ---- 4a: the by-input table (coalesced over the rows)
    if (!(EC_KNOCK & 4)) {
        for (int jl = tid; jl < n; jl += EC_THREADS) {
            const long long j = (long long)r0 + jl;
            if (CACHE && jl == tid) {
#pragma unroll
                for (int t = 0; t < NC; ++t)
                    if (t < nslots) {
                        const int il = il_mine[t];
                        int v = -1;
                        if (il >= 0 && il < mw) v = PACKED ? ((((int)base + il) << 3) | mine.o_star) : (int)base + il;
                        nbr_out[(long long)t * N + j] = v;
                    }
            } else {
                unsigned done = 0u;
                visit(jl, [&](int t, int cell, int k) {
                    const int il = (int)cellid[cell];
                    int v = -1;
                    if (il < mw) v = PACKED ? ((((int)base + il) << 3) | (k - t * g.kl)) : (int)base + il;
                    nbr_out[(long long)t * N + j] = v;
                    done |= 1u << t;
                });
                for (int t = 0; t < nslots; ++t)
                    if (!((done >> t) & 1u)) nbr_out[(long long)t * N + j] = -1;
            }
        }
        // ---- 4b: coordinates of the event's sites, one thread per CELL of the event's grid (the cell is the coordinate)
        for (int c = tid; c < g.cells_e; c += EC_THREADS) {
            const int il = (int)cellid[c];
            if (il >= mw) continue;                     // 0xFFFF: not a site; beyond the capacity: dropped
            unsigned rem = (unsigned)c;
            int *o = out_idx + (base + il) * cols;
            o[0] = e;
#pragma unroll
            for (int d = 3; d >= 0; --d)
                if (d <= last) {
                    if (g.oshape[d] > 1) {
                        const unsigned qd = __umulhi(rem, g.dmagic[d]);
                        o[1 + d] = (int)(rem - qd * (unsigned)g.oshape[d]);
                        rem = qd;
                    } else {
                        o[1 + d] = 0;
                    }
                }
        }
    }
    // ---- 4c: the image passes written out (the first one was built above, beside the look-back)
    if (nbr_in && M_e > 0) {
        for (int k0 = 0; k0 < g.K; k0 += gs) {
            const int k1 = k0 + gs < g.K ? k0 + gs : g.K;
            if (k0 > 0) {
                lds_barrier();                        // the previous pass has been written out
                image_pass(k0, k1);
            }
            for (int k = k0 + wid; k < k1; k += EC_WAVES) {
                const unsigned short *src = img + (k - k0) * mstride;
                int *dst = nbr_in + (long long)k * M_cap + base;
                for (int i = lane; i < mw; i += 64) {
                    const unsigned v = src[i];
                    dst[i] = v == 0xFFFFu ? -1 : r0 + (int)v;
                }
            }
        }
    }
    if (!structured || f_fail) {
        // the event table is meaningless (or this event is beyond the tables): no row has an output -- the consumers of
        // the by-input table are bounded by the INPUT row count, so every row they can reach must say "none"
        long long j_lo, j_hi;
        if (!structured) {                              // the workgroups share ALL the rows out evenly
            const long long per = (Nv + B - 1) / B;
            j_lo = (long long)e * per;
            j_hi = j_lo + per < Nv ? j_lo + per : Nv;
        } else {
            j_lo = r0;
            j_hi = r1;
        }
        for (int t = 0; t < nslots; ++t)
            for (long long j = j_lo + tid; j < j_hi; j += EC_THREADS) nbr_out[(long long)t * N + j] = -1;
    }
    if (cell_row) {
        int *dst = cell_row + (long long)e * g.cells_e;
        for (int c = tid; c < g.cells_e; c += EC_THREADS) {
            const int il = (int)cellid[c];
            dst[c] = (n > 0 && il < mw) ? (int)(base + il) : -1;          // il = 0xFFFF: no site
        }
    }
    f_range = __syncthreads_or(f_range);
    timed_out = __syncthreads_or(timed_out);
    if (tid == 0) {
        ev_out[e] = (int)(base < M_cap ? base : M_cap);
        // STICKY: only ever set here (a captured step is checked every so many replays)
        if (!structured || timed_out || f_fail) flags[0] = 1;
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
    for (int d = 0; d < 4; ++d) {
        G->oshape[d] = d < g->ndim ? g->out_shape[d] : 1;
        // floor(c / s) == umulhi(c, floor(2^32 / s) + 1) for c < 2^16, s < 2^16
        G->dmagic[d] = G->oshape[d] > 1 ? (unsigned)((1ull << 32) / (unsigned long long)G->oshape[d]) + 1u : 0u;
    }
    return true;
}

// candidate slots of the kernel's register cache (see Cand): packed form one per leading offset, dense form one per offset
void fill_digits(const wfs_geometry *g, ECGeo *G, bool packed) {
    const int last = g->ndim - 1;
    const int nslots = packed ? G->Kq : g->K;
    for (int t = 0; t < 32; ++t) {
        G->dig[t] = 0;
        if (t >= nslots) continue;
        int rem = packed ? t : t / G->kl;
        unsigned dg = packed ? 0u : (unsigned)(t % G->kl) << 24;
        for (int d = last - 1; d >= 0; --d) {
            dg |= (unsigned)(rem % g->ksize[d]) << (8 * d);
            rem /= g->ksize[d];
        }
        G->dig[t] = dg;
    }
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
    return 64 + (size_t)EC_REPLICAS * ec_replica_stride(batch_size > 0 ? batch_size : 1);
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
    // LDS: the tickets (4 B per cell) and, in their place once the ids exist, the image of the by-output table's rows of
    // one event ([offsets of a pass][M_e] uint16); two uint16 maps per cell behind them.  Small grids get a larger image
    // (up to 24 KB) so that a pass covers most offsets
    long long img = (long long)G.cells_e * 4;
    long long want = (long long)g->K * (((long long)G.cells_e + 1) & ~1ll) * 2;
    if (want > 24 * 1024) want = 24 * 1024;
    if (img < want) img = want;
    img = (img + 15) / 16 * 16;
    const size_t lds = (size_t)img + (size_t)G.cells_e * 2;
    unsigned *st = (unsigned *)state;
    unsigned long long *pub = (unsigned long long *)((char *)state + 64);
    const dim3 grid((unsigned)B);
    const bool packed = packed_kl != 0;
    fill_digits(g, &G, packed);
    const int nslots = packed ? G.Kq : g->K;
#define WFS_EC(P, NQ, TH)                                                                                                \
    do {                                                                                                                 \
        static bool attr = false;                                                                                        \
        if (!attr) {                                                                                                     \
            WFS_HIP_CHECK(hipFuncSetAttribute((const void *)k_ev_conv<P, NQ, TH>,                                        \
                                              hipFuncAttributeMaxDynamicSharedMemorySize,                                \
                                              158 * 1024));                                                              \
            attr = true;                                                                                                 \
        }                                                                                                                \
        k_ev_conv<P, NQ, TH><<<grid, dim3(TH), lds, stream>>>(G, (int)img, indices, N, (const long long *)n_dev, events_in, B, \
                                                       M_cap, out_indices, (long long *)m_dev, events_out, nbr_out,      \
                                                       nbr_in, cell_row, overflow_dev, flags, pub, st);                   \
    } while (0)
    // threads per workgroup: 512.  Alone, at the PSD batch (one event per CU), 1024 threads are faster (31 vs 45 us: the
    // launch lasts as long as its largest event, one row per thread and phase) -- but inside a training step the builds
    // run on a side branch beside the first layers with time to spare, and what counts is the room they leave those
    // layers' kernels on every CU (EC_MIN_WAVES above); beyond 256 events throughput counts: three 512-thread workgroups
    // to a CU (128 vs 174 us at 2048 events).  WFS_EC_THREADS = 256 / 512 / 1024 overrides (experiments)
    static const int ec_forced = [] {
        const char *e = getenv("WFS_EC_THREADS");
        const int v = e ? atoi(e) : 0;
        return v == 256 || v == 512 || v == 1024 ? v : 0;
    }();
    const int ec_threads = ec_forced ? ec_forced : 512;
    if (packed && nslots <= 9 && ec_threads == 1024) WFS_EC(true, 9, 1024);
    else if (packed && nslots <= 9 && ec_threads == 256) WFS_EC(true, 9, 256);
    else if (packed && nslots <= 9) WFS_EC(true, 9, 512);
    else if (packed) WFS_EC(true, 32, 512);
    else if (nslots <= 9) WFS_EC(false, 9, 512);
    else WFS_EC(false, 32, 512);
#undef WFS_EC
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
