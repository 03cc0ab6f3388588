// rulebook_chain.hip -- the rulebooks of a whole stack of sparse-conv layers in TWO launches, one workgroup per event.
//
// Replaces, for a SparseSequential's conv layers, the per-layer torch.ops.spconv.get_indice_pairs calls of spconv 1.2.1
// (reference call sites src/models/SPConvBlocks.py:75,134,498; the C2 net of config/psd_c2_3d.json is SubM x3 on one
// key + two strided SparseConv3d).  rulebook.hip builds ONE layer with 3 (SubM) or 6 (regular) chip-wide launches whose
// phases are separated by kernel boundaries (site table -> tickets -> first-ticket masks -> row scan -> ids -> tables);
// at the PSD batch sizes that is 17 launches and a quarter of the step's kernel time.  Rulebooks never cross events
// (the site key contains the batch index) and an event is a few hundred voxels, so here ONE WORKGROUP
// builds everything for ONE EVENT with its site tables in LDS and __syncthreads() between the phases:
//
//   count kernel  per event: row range [start, start + n) of the event (cooperative search in the batch column), and
//                 for every regular layer the number of output sites it opens (site SETS only);
//   build kernel  per event: exclusive sums of those counts = where the event's output rows start in every layer, then
//                 SubM neighbour tables, first-seen output numbering (A.3: rows in order, offsets in order), nbr_out /
//                 nbr_in / out_indices of every layer and the cell -> row map of layers that ask for one.
//
// Bit-identical to the sequential CPU algorithm PROVIDED the batch column is non-decreasing (events contiguous and in
// order -- what the reference's collate_fn delivers, src/engineering/PSDDataModule.py:10-20: global first-seen order is
// then event order followed by the event's own first-seen order).  The kernels verify it (every row of an event's range
// carries its batch id; the ranges add up to N) and raise the chain's error flag otherwise, as they do for an event too
// large for the LDS tables (CH_SITES rows in any of its site sets); callers then use the per-layer builds.
// Duplicate coordinates follow A.3 as in rulebook.hip (SubM: last row wins; regular: tickets).
#include <stdlib.h>

#include "wfs_common.h"

namespace {

constexpr int CH_THREADS = 1024;
constexpr int CH_SITES = 2048;            // rows of one event in any site set (two row chunks of CH_THREADS)
constexpr int CH_HASH = 4096;             // hash slots per table (load <= 0.5)
constexpr int CH_GRID = 8192;             // an event's volume up to this uses a direct grid instead
constexpr int CH_TAB_WORDS = 8192;        // 32 KiB per table: direct [value] x cells, hash [key | value] x CH_HASH
constexpr unsigned EMPTY = 0xFFFFFFFFu;
static_assert(2 * CH_HASH <= CH_TAB_WORDS && CH_GRID <= CH_TAB_WORDS, "table layouts must fit the table memory");
constexpr int CH_CACHE_ITEMS = 36864;     // (row, offset) candidates whose table slot is parked in LDS (16 bits each)
// LDS: two site lists (coordinates packed 16 bits per dim, 8 B per site), first-ticket masks, one table, the slot cache
constexpr size_t CH_LDS_BYTES = (size_t)(4 * CH_SITES + CH_SITES + CH_TAB_WORDS) * 4 + (size_t)CH_CACHE_ITEMS * 2;

// Geometry of one layer, dims RIGHT-ALIGNED to four levels (level 3 = last = fastest dim; missing leading dims are
// size-1 dummies), so that one four-level loop nest serves every ndim.
struct Geo {
    int ndim, K;
    int spatial[4], out_shape[4], ksize[4], stride[4], padding[4], dilation[4];
    int shift[4];                         // log2(stride) if it is a power of two, else -1
    float inv_stride[4];
    int in_volume, out_volume;
};

struct Layer {
    Geo g;
    int subm;
    int in_direct, out_direct;            // table kinds (by volume)
    int *nbr_out, *nbr_in, *out_indices;
    long long N_cap, M_cap;
    long long *m_dev;
    int *overflow_dev;
    unsigned *cell_ticket;
    int *cell_row;
};

struct Chain {
    int nlayers, batch;
    Layer L[WFS_CHAIN_MAX_LAYERS];
    const int *idx;
    long long N;
    const long long *n_dev;
    int *counts;                          // [batch][2 + WFS_CHAIN_MAX_LAYERS]: start, n, outputs of layer l
    int debug;                            // timing experiments: stop after stage `debug` (0 = run everything)
    int *flags;                           // [0] error bits (1: rows not grouped by event / bad index, 2: event too large)
};
constexpr int CW = 2 + WFS_CHAIN_MAX_LAYERS;

// A layer's geometry as the loops use it: read ONCE per layer from the kernel arguments and pinned in scalar registers.
// (Left to itself hipcc re-loads every field from the argument segment at every use inside the loops: a dozen dependent
// s_load + s_waitcnt lgkmcnt(0) per iteration, each also draining the LDS queue.)
#define WFS_PIN(x) asm volatile("" : "+s"(x))
struct LGeo {
    int K;
    int spatial[4], out_shape[4], ksize[4], stride[4], padding[4], dilation[4], shift[4];
    float inv_stride[4];
    __device__ __forceinline__ void load(const Geo &g) {
        K = g.K;
        WFS_PIN(K);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            spatial[d] = g.spatial[d];
            out_shape[d] = g.out_shape[d];
            ksize[d] = g.ksize[d];
            stride[d] = g.stride[d];
            padding[d] = g.padding[d];
            dilation[d] = g.dilation[d];
            shift[d] = g.shift[d];
            inv_stride[d] = g.inv_stride[d];
            WFS_PIN(spatial[d]);
            WFS_PIN(out_shape[d]);
            WFS_PIN(ksize[d]);
            WFS_PIN(stride[d]);
            WFS_PIN(padding[d]);
            WFS_PIN(dilation[d]);
            WFS_PIN(shift[d]);
            WFS_PIN(inv_stride[d]);
        }
    }
};

// ---- site table in LDS: one value word per slot.  Output table: EMPTY -> smallest ticket (atomicMin) -> local id;
// site table of a row set: -1 -> row of the event (atomicMax: the last duplicate wins).  Hash mode adds a key word.
struct Tab {
    int *w;         // CH_TAB_WORDS words
    int direct;
    __device__ __forceinline__ int *val() const { return w + (direct ? 0 : CH_HASH); }
    __device__ void clear(int vol) const {
        const int n = direct ? vol : 2 * CH_HASH;
        for (int i = threadIdx.x; i < n; i += CH_THREADS) w[i] = -1;
    }
    __device__ __forceinline__ int insert(int key) const {
        if (direct) return key;
        unsigned s = ((unsigned)key * 0x9E3779B1u) >> 20;                // 32 - log2(CH_HASH)
        while (true) {
            int prev = atomicCAS(&w[s], -1, key);
            if (prev == -1 || prev == key) return (int)s;
            s = (s + 1) & (CH_HASH - 1);
        }
    }
    __device__ __forceinline__ int find(int key) const {                 // slot or -1
        if (direct) return key;
        unsigned s = ((unsigned)key * 0x9E3779B1u) >> 20;
        while (true) {
            int cur = w[s];
            if (cur == key) return (int)s;
            if (cur == -1) return -1;
            s = (s + 1) & (CH_HASH - 1);
        }
    }
    __device__ __forceinline__ int get(int key) const {                  // value of a key, -1 if absent
        const int s = find(key);
        return s >= 0 ? val()[s] : -1;
    }
};
static_assert(CH_HASH == 4096, "hash shift above assumes 4096 slots");

// coordinates of a site: levels 0,1 in `lo`, levels 2,3 in `hi`, 16 bits each
struct Site {
    unsigned lo, hi;
};
__device__ __forceinline__ int site_x(const Site &s, int d) {
    return (int)(((d < 2 ? s.lo : s.hi) >> (16 * (d & 1))) & 0xFFFFu);
}
__device__ __forceinline__ void site_put(Site &s, int d, int x) {
    if (d < 2) s.lo |= (unsigned)x << (16 * (d & 1)); else s.hi |= (unsigned)x << (16 * (d & 1));
}

// t / stride for 0 <= t < 2^16 with full-rate instructions (v_mul_lo / v_mul_hi are quarter rate)
__device__ __forceinline__ int div_stride(const LGeo &g, int d, int t) {
    if (g.stride[d] == 1) return t;
    if (g.shift[d] >= 0) return t >> g.shift[d];
    int q = (int)(((float)t + 0.5f) * g.inv_stride[d]);
    if (__mul24(q, g.stride[d]) > t) --q;
    if (__mul24(q + 1, g.stride[d]) <= t) ++q;
    return q;
}

// One level (dim) of a candidate: input coordinate x, kernel offset j -> output coordinate (A.3 getValidOutPos:
// x + p - j * d = o * s).  Scalar (uniform) branches pick the cheap form for stride 1 / power-of-two strides.
__device__ __forceinline__ bool out_level(const LGeo &g, int d, int x, int j, int *o) {
    const int t = x + g.padding[d] - __mul24(j, g.dilation[d]);
    if (g.stride[d] == 1) {
        *o = t;
        return (unsigned)t < (unsigned)g.out_shape[d];
    }
    if (g.shift[d] >= 0) {
        *o = t >> g.shift[d];
        return t >= 0 && (t & (g.stride[d] - 1)) == 0 && *o < g.out_shape[d];
    }
    const int tc = t < 0 ? 0 : t;
    *o = div_stride(g, d, tc);
    return t >= 0 && __mul24(*o, g.stride[d]) == t && *o < g.out_shape[d];
}
// A PLANE of candidates = all offsets of the last level for fixed offsets of the levels before it.  The levels before
// the last are evaluated once per plane (dummy leading levels of a < 4-dim geometry are skipped: scalar branch):
// partial key, validity, partial coordinates.
struct PlaneHead {
    int lin;
    bool ok;
    Site so;
};
__device__ __forceinline__ PlaneHead plane_head(const LGeo &g, int nd, unsigned offs, const Site &x) {
    PlaneHead h = {0, true, {0u, 0u}};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (d < 4 - nd) continue;
        int o;
        const bool okd = out_level(g, d, site_x(x, d), (int)((offs >> (8 * d)) & 0xFFu), &o);
        h.ok = h.ok && okd;
        h.lin = __mul24(h.lin, g.out_shape[d]) + o;
        site_put(h.so, d, o & 0xFFFF);
    }
    return h;
}
// the candidate of plane head h with last-level offset j3: key or -1, coordinates in *os
__device__ __forceinline__ int plane_candidate(const LGeo &g, const PlaneHead &h, int x3, int j3, Site *os) {
    int o;
    const bool ok = out_level(g, 3, x3, j3, &o) && h.ok;
    *os = h.so;
    site_put(*os, 3, o & 0xFFFF);
    return ok ? __mul24(h.lin, g.out_shape[3]) + o : -1;
}
// a single candidate (the rare paths: numbering, events beyond the slot cache)
__device__ __forceinline__ int out_site(const LGeo &g, int nd, unsigned offs, const Site &x, Site *os) {
    const PlaneHead h = plane_head(g, nd, offs, x);
    return plane_candidate(g, h, site_x(x, 3), (int)(offs >> 24), os);
}
__device__ __forceinline__ unsigned pack_offsets(const Geo &g, int k) {
    unsigned p = 0;
    for (int d = 3; d >= 0; --d) {
        p |= (unsigned)(k % g.ksize[d]) << (8 * d);
        k /= g.ksize[d];
    }
    return p;
}
// (k, r) of flat item i = k * n + r without an integer division: exact for i < 2^22 (here i < 2^16)
__device__ __forceinline__ void split_item(int i, int n, float inv_n, int *k, int *r) {
    int q = (int)(((float)i + 0.5f) * inv_n);
    int rem = i - __mul24(q, n);
    if (rem < 0) {
        --q;
        rem += n;
    } else if (rem >= n) {
        ++q;
        rem -= n;
    }
    *k = q;
    *r = rem;
}

__device__ __forceinline__ int lin_key(const int *shape, const Site &s) {
    int lin = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) lin = __mul24(lin, shape[d]) + site_x(s, d);
    return lin;
}

__device__ __forceinline__ long long valid_rows(long long N, const long long *n_dev) {
    long long v = n_dev ? *n_dev : N;
    return v < N ? v : N;
}

// block-wide exclusive scan of one int per thread (CH_THREADS threads); *total = block sum
__device__ __forceinline__ int block_excl_scan(int v, int *total, int *wsum /* [16] LDS */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int n = __shfl_up(inc, d, 64);
        if (lane >= d) inc += n;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < CH_THREADS / 64; ++w) {
        int s = wsum[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// first row p in [lo, hi) whose batch id is >= e (hi if none), by all CH_THREADS threads: 1024-ary search, two
// dependent reads for a million rows instead of twenty
__device__ long long coop_lower_bound(const int *idx, int stride, long long lo, long long hi, int e, int *sF) {
    while (hi > lo) {
        const long long len = hi - lo;
        const long long step = (len + CH_THREADS - 1) / CH_THREADS;
        if (threadIdx.x == 0) *sF = CH_THREADS;
        __syncthreads();
        const long long p = lo + (long long)threadIdx.x * step;
        if (p < hi && idx[p * stride] >= e) atomicMin(sF, (int)threadIdx.x);
        __syncthreads();
        const int F = *sF;
        __syncthreads();
        // the answer lies in (p_{F-1}, p_F]
        const long long nlo = F == 0 ? lo : lo + (long long)(F - 1) * step + 1;
        const long long nhi = F == CH_THREADS ? hi : (lo + (long long)F * step < hi ? lo + (long long)F * step : hi);
        if (step == 1) return nhi;
        lo = nlo;
        hi = nhi;
    }
    return hi;
}

// Work is (row, offset)-parallel: item i = k * n + r, so that all 16 waves of the block have LDS operations in flight
// (a thread-per-row form leaves a 300-row event with 5 busy waves whose 27 dependent LDS round trips per pass run back
// to back: measured 143 us for the build kernel).  The one expensive step, a candidate's output site (out_site, ~50
// instructions + the table insert), runs ONCE per item: its slot is parked in LDS as 16 bits and the later passes
// (first tickets, tables) just read it.
template <bool COUNT>
__global__ void __launch_bounds__(CH_THREADS) k_chain(Chain c) {
    extern __shared__ __attribute__((aligned(16))) int lds[];
    __shared__ int sMisc[8];
    __shared__ int sW[CH_THREADS / 64];
    __shared__ unsigned sOff[32];
    __shared__ long long sRed[2 * WFS_CHAIN_MAX_LAYERS + 2];
    Site *siteA = (Site *)lds, *siteB = (Site *)lds + CH_SITES;
    unsigned *rowmask = (unsigned *)(lds + 4 * CH_SITES);
    Tab tab = {lds + 5 * CH_SITES, 0};
    unsigned short *slotc = (unsigned short *)(lds + 5 * CH_SITES + CH_TAB_WORDS);
    const int e = blockIdx.x;
    const int tid = threadIdx.x;
    const int nd = c.L[0].g.ndim, stride = nd + 1;
    const long long nv = valid_rows(c.N, c.n_dev);
    int *cnt = c.counts + (long long)e * CW;

    long long start;
    int n;
    long long base[WFS_CHAIN_MAX_LAYERS];
#pragma unroll
    for (int l = 0; l < WFS_CHAIN_MAX_LAYERS; ++l) base[l] = 0;
    if (COUNT) {
        start = coop_lower_bound(c.idx, stride, 0, nv, e, &sMisc[0]);
        // the event's rows end within CH_SITES of the start, or the event is too large for the tables anyway
        const long long lim = start + CH_SITES + 1 < nv ? start + CH_SITES + 1 : nv;
        const long long end = coop_lower_bound(c.idx, stride, start, lim, e + 1, &sMisc[0]);
        n = (int)(end - start);
        if (n > CH_SITES) {
            if (tid == 0) atomicOr(&c.flags[0], 2);
            n = CH_SITES;
        }
        if (tid == 0) {
            cnt[0] = (int)start;
            cnt[1] = n;
        }
    } else {
        // where this event's rows start in every regular layer's output set: sums over the events in front of it
        long long pre[WFS_CHAIN_MAX_LAYERS], tot[WFS_CHAIN_MAX_LAYERS], nsum = 0;
#pragma unroll
        for (int l = 0; l < WFS_CHAIN_MAX_LAYERS; ++l) pre[l] = tot[l] = 0;
        for (int b = tid; b < c.batch; b += CH_THREADS) {
            const int *cb = c.counts + (long long)b * CW;
            nsum += cb[1];
#pragma unroll
            for (int l = 0; l < WFS_CHAIN_MAX_LAYERS; ++l) {
                const int v = l < c.nlayers ? cb[2 + l] : 0;
                tot[l] += v;
                if (b < e) pre[l] += v;
            }
        }
        if (tid < 2 * WFS_CHAIN_MAX_LAYERS + 2) sRed[tid] = 0;
        __syncthreads();
        typedef unsigned long long u64;
#pragma unroll
        for (int l = 0; l < WFS_CHAIN_MAX_LAYERS; ++l) {
            long long a = pre[l], t = tot[l];       // wave-level sums first, one LDS atomic per wave and quantity
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                a += __shfl_xor(a, d, 64);
                t += __shfl_xor(t, d, 64);
            }
            if ((tid & 63) == 0) {
                atomicAdd((u64 *)&sRed[2 * l], (u64)a);
                atomicAdd((u64 *)&sRed[2 * l + 1], (u64)t);
            }
        }
        {
            long long sN = nsum;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) sN += __shfl_xor(sN, d, 64);
            if ((tid & 63) == 0) atomicAdd((u64 *)&sRed[2 * WFS_CHAIN_MAX_LAYERS], (u64)sN);
        }
        __syncthreads();
#pragma unroll
        for (int l = 0; l < WFS_CHAIN_MAX_LAYERS; ++l) base[l] = sRed[2 * l];
        if (e == 0 && tid == 0) {
            int bad = c.flags[0];
            if (sRed[2 * WFS_CHAIN_MAX_LAYERS] != nv) {          // the per-event ranges do not tile the rows: not grouped
                bad |= 1;
                atomicOr(&c.flags[0], 1);
            }
            for (int l = 0; l < c.nlayers; ++l) {
                if (c.L[l].subm) continue;
                const long long M = sRed[2 * l + 1];
                if (c.L[l].m_dev) *c.L[l].m_dev = M < c.L[l].M_cap ? M : c.L[l].M_cap;
                if (c.L[l].overflow_dev) *c.L[l].overflow_dev = (M > c.L[l].M_cap || bad) ? 1 : 0;
            }
        }
        start = cnt[0];
        n = cnt[1];
    }

    if (c.debug == 1) return;
    // ---- the event's input rows, coordinates packed (levels 4 - ndim .. 3)
    {
        LGeo g;
        g.load(c.L[0].g);
        bool bad = false;
        for (int r = tid; r < n; r += CH_THREADS) {
            const int *row = c.idx + (start + r) * stride;
            bad = bad || row[0] != e;
            Site xs = {0u, 0u};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int col = d - (4 - nd);
                if (col < 0) continue;
                const int x = row[1 + col];
                bad = bad || x < 0 || x >= g.spatial[d];
                site_put(xs, d, x < 0 ? 0 : (x >= g.spatial[d] ? g.spatial[d] - 1 : x));
            }
            siteA[r] = xs;
        }
        if (COUNT && bad) atomicOr(&c.flags[0], 1);
    }
    long long in_base = start;
    bool tab_is_site_table = false;       // does `tab` hold key -> row for the CURRENT row set?
    __syncthreads();
    if (c.debug == 2) return;

    for (int l = 0; l < c.nlayers; ++l) {
        const Layer &L = c.L[l];
        LGeo g;
        g.load(L.g);
        const int K = g.K;
        // the layer's pointers / capacities likewise once, in registers
        int *const p_nbr_out = L.nbr_out, *const p_nbr_in = L.nbr_in, *const p_out_indices = L.out_indices;
        const long long N_cap = L.N_cap, M_cap = L.M_cap;
        const int in_volume = L.g.in_volume, out_volume = L.g.out_volume;
        const bool is_subm = L.subm != 0;
        const int items = n * K;
        const int ks3 = g.ksize[3];                 // offsets of the last level = candidates per plane
        const int pitems = n * (K / ks3);           // (plane, row) work items
        const float inv_n = n > 0 ? 1.0f / (float)n : 0.f;
        if (tid < K) sOff[tid] = pack_offsets(L.g, tid);
        if (c.debug == 3 + l) return;
        if (is_subm) {
            if (COUNT) {
                __syncthreads();           // sOff is rewritten by the next layer
                continue;
            }
            if (!tab_is_site_table) {
                // site table of the row set: key -> row of the event (duplicates: the last row wins, A.3)
                tab.direct = L.in_direct;
                tab.clear(in_volume);
                __syncthreads();
                for (int r = tid; r < n; r += CH_THREADS) atomicMax(&tab.val()[tab.insert(lin_key(g.spatial, siteA[r]))], r);
                tab_is_site_table = true;
            }
            __syncthreads();
            for (int i = tid; i < pitems; i += CH_THREADS) {
                int pl, r;
                split_item(i, n, inv_n, &pl, &r);
                const Site x = siteA[r];
                const int k0 = __mul24(pl, ks3), x3 = site_x(x, 3);
                const PlaneHead h = plane_head(g, nd, sOff[k0], x);
                int *dst = p_nbr_out + (long long)k0 * N_cap + in_base + r;
                for (int j3 = 0; j3 < ks3; ++j3) {
                    Site os;
                    const int key = plane_candidate(g, h, x3, j3, &os);
                    int res = -1;
                    if (key >= 0) {
                        const int rr = tab.get(key);
                        if (rr >= 0) res = (int)(in_base + rr);
                    }
                    dst[(long long)j3 * N_cap] = res;
                }
            }
            __syncthreads();
            continue;          // the site set is unchanged
        }
        // ---- regular conv: output site table
        const bool cached = items <= CH_CACHE_ITEMS;
        tab.direct = L.out_direct;
        tab.clear(out_volume);
        for (int r = tid; r < n; r += CH_THREADS) rowmask[r] = 0u;
        if (tid == 0) sMisc[1] = 0;
        __syncthreads();
        // the slot of candidate i (cache, or evaluated again for an event too large for the cache)
        auto slot_of = [&](int i, int k, int r) -> int {
            if (cached) {
                const int sl = slotc[i];
                return sl == 0xFFFF ? -1 : sl;
            }
            Site os;
            const int key = out_site(g, nd, sOff[k], siteA[r], &os);
            return key >= 0 ? tab.find(key) : -1;
        };
        // A: tickets -- every candidate offers (row * K + offset) to its output site, the smallest one opens the site
        for (int i = tid; i < pitems; i += CH_THREADS) {
            int pl, r;
            split_item(i, n, inv_n, &pl, &r);
            const Site x = siteA[r];
            const int k0 = __mul24(pl, ks3), x3 = site_x(x, 3);
            const PlaneHead h = plane_head(g, nd, sOff[k0], x);
            for (int j3 = 0; j3 < ks3; ++j3) {
                const int k = k0 + j3;
                Site os;
                const int key = plane_candidate(g, h, x3, j3, &os);
                int sl = -1;
                if (key >= 0) {
                    sl = tab.insert(key);
                    const unsigned old = atomicMin((unsigned *)&tab.val()[sl], (unsigned)(__mul24(r, K) + k));
                    if (COUNT && old == EMPTY) {             // a new site: append it (any order) for the next layer
                        const int pos = atomicAdd(&sMisc[1], 1);
                        if (pos < CH_SITES) siteB[pos] = os;
                    }
                }
                if (!COUNT && cached) slotc[__mul24(k, n) + r] = (unsigned short)(sl < 0 ? 0xFFFF : sl);
            }
        }
        __syncthreads();
        int m;
        if (COUNT) {
            m = sMisc[1];
            if (m > CH_SITES) {
                if (tid == 0) atomicOr(&c.flags[0], 2);
                m = CH_SITES;
            }
            if (tid == 0) cnt[2 + l] = m;
        } else {
            // B: first tickets -- bit k of rowmask[r] = candidate (r, k) opened its site
            for (int i = tid; i < items; i += CH_THREADS) {
                int k, r;
                split_item(i, n, inv_n, &k, &r);
                const int sl = slot_of(i, k, r);
                if (sl >= 0 && (unsigned)tab.val()[sl] == (unsigned)(r * K + k)) atomicOr(&rowmask[r], 1u << k);
            }
            __syncthreads();
            // C: first-seen numbering -- rows in order, offsets in order (A.3) = exclusive scan of the first-ticket
            // counts; the ids replace the tickets in the table (no ticket is read after pass B)
            int carry = 0;
            for (int r0 = 0; r0 < n; r0 += CH_THREADS) {
                const int r = r0 + tid;
                unsigned mask = r < n ? rowmask[r] : 0u;
                int tot;
                int ex = block_excl_scan(__popc(mask), &tot, sW) + carry;
                while (mask) {
                    const int k = __builtin_ctz(mask);
                    mask &= mask - 1;
                    Site os;
                    const int key = out_site(g, nd, sOff[k], siteA[r], &os);
                    tab.val()[tab.find(key)] = ex;
                    if (ex < CH_SITES) siteB[ex] = os;
                    ++ex;
                }
                carry += tot;
            }
            __syncthreads();
            m = carry < CH_SITES ? carry : CH_SITES;          // carry > CH_SITES was flagged by the count kernel
            const long long out_base = base[l];
            // nbr_in of the event's output rows starts as "no input"; out_indices
            {
                const int oitems = m * K;
                const float inv_m = m > 0 ? 1.0f / (float)m : 0.f;
                for (int i = tid; i < oitems; i += CH_THREADS) {
                    int k, o;
                    split_item(i, m, inv_m, &k, &o);
                    const long long gid = out_base + o;
                    if (gid < M_cap) p_nbr_in[(long long)k * M_cap + gid] = -1;
                }
                for (int o = tid; o < m; o += CH_THREADS) {
                    const long long gid = out_base + o;
                    if (gid >= M_cap) continue;
                    const Site os = siteB[o];
                    int *oi = p_out_indices + gid * (nd + 1);
                    oi[0] = e;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int col = d - (4 - nd);
                        if (col >= 0) oi[1 + col] = site_x(os, d);
                    }
                }
            }
            __syncthreads();          // the fills above have completed (vmcnt(0) + barrier) before the entries below land
            // D: the tables -- nbr_out[k][input row] = output row, nbr_in[k][output row] = input row (sites are distinct
            // by the caller's contract; with duplicate coordinates the larger row would have to win, A.3)
            for (int i = tid; i < items; i += CH_THREADS) {
                int k, r;
                split_item(i, n, inv_n, &k, &r);
                const int sl = slot_of(i, k, r);
                int res = -1;
                if (sl >= 0) {
                    const long long gid = out_base + tab.val()[sl];
                    if (gid < M_cap) {
                        res = (int)gid;
                        p_nbr_in[(long long)k * M_cap + gid] = (int)(in_base + r);
                    }
                }
                p_nbr_out[(long long)k * N_cap + in_base + r] = res;
            }
            if (L.cell_row) {
                unsigned *const p_ct = L.cell_ticket;
                int *const p_cr = L.cell_row;
                for (int cell = tid; cell < out_volume; cell += CH_THREADS) {
                    const int o = tab.get(cell);
                    const long long gid = o >= 0 ? out_base + o : -1;
                    const bool ok = gid >= 0 && gid < M_cap;
                    p_ct[(long long)e * out_volume + cell] = ok ? 0u : EMPTY;
                    p_cr[(long long)e * out_volume + cell] = ok ? (int)gid : -1;
                }
            }
            in_base = out_base;
        }
        __syncthreads();
        // the outputs are the next layer's inputs; the table (key -> local id) is their site table
        Site *t = siteA;
        siteA = siteB;
        siteB = t;
        n = m;
        tab_is_site_table = !COUNT;
    }
}

bool geo_from(const wfs_geometry *g, Geo *G) {
    G->ndim = g->ndim;
    G->K = g->K;
    long long iv = 1, ov = 1;
    const int lead = 4 - g->ndim;                    // dims are right-aligned to four levels
    for (int lv = 0; lv < 4; ++lv) {
        const int i = lv - lead;
        const bool real = i >= 0;
        G->spatial[lv] = real ? g->spatial[i] : 1;
        G->out_shape[lv] = real ? g->out_shape[i] : 1;
        G->ksize[lv] = real ? g->ksize[i] : 1;
        G->stride[lv] = real ? g->stride[i] : 1;
        G->padding[lv] = real ? g->padding[i] : 0;
        G->dilation[lv] = real ? g->dilation[i] : 1;
        const int sd = G->stride[lv] > 0 ? G->stride[lv] : 1;
        G->shift[lv] = -1;
        for (int b = 0; b < 16; ++b)
            if ((1 << b) == sd) G->shift[lv] = b;
        G->inv_stride[lv] = 1.0f / (float)sd;
        if (real) {
            iv *= g->spatial[i];
            ov *= g->out_shape[i];
            // coordinates travel packed 16 bits per dim and the index arithmetic is 24-bit
            if (g->spatial[i] > 32767 || g->out_shape[i] > 32767 || g->padding[i] > 32767 || g->stride[i] > 4096 ||
                g->dilation[i] > 4096)
                return false;
        }
    }
    if (iv >= (1ll << 24) || ov >= (1ll << 24)) return false;
    G->in_volume = (int)iv;
    G->out_volume = (int)ov;
    return true;
}

int fill_chain(const wfs_chain_layer *layers, int nlayers, const int32_t *indices, int64_t N, const int64_t *n_dev,
               void *workspace, size_t workspace_bytes, Chain *c) {
    WFS_REQUIRE(layers && nlayers >= 1 && nlayers <= WFS_CHAIN_MAX_LAYERS, WFS_EINVAL, "1..%d layers per chain",
                WFS_CHAIN_MAX_LAYERS);
    const int batch = layers[0].geo.batch_size;
    WFS_REQUIRE(batch >= 1 && batch <= 65535, WFS_EINVAL, "batch_size %d out of range for the event-parallel build", batch);
    WFS_REQUIRE(N >= 0 && N < (1ll << 31), WFS_EINVAL, "N out of range");
    WFS_REQUIRE(workspace && workspace_bytes >= wfs_rulebook_chain_workspace_bytes(batch), WFS_EWORKSPACE,
                "chain workspace too small");
    c->nlayers = nlayers;
    c->batch = batch;
    c->idx = indices;
    c->N = N;
    c->n_dev = (const long long *)n_dev;
    c->flags = (int *)workspace;
    c->debug = getenv("WFS_CHAIN_DEBUG") ? atoi(getenv("WFS_CHAIN_DEBUG")) : 0;
    c->counts = (int *)workspace + 64;
    const wfs_geometry *prev = nullptr;
    for (int l = 0; l < nlayers; ++l) {
        const wfs_chain_layer &s = layers[l];
        Layer &L = c->L[l];
        WFS_REQUIRE(s.geo.K >= 1 && s.geo.K <= 32, WFS_EINVAL, "layer %d: 1 <= K <= 32 (got %d)", l, s.geo.K);
        WFS_REQUIRE(s.geo.batch_size == batch && s.geo.ndim == layers[0].geo.ndim, WFS_EINVAL,
                    "layer %d: batch size / ndim differ from layer 0", l);
        WFS_REQUIRE(geo_from(&s.geo, &L.g), WFS_EOVERFLOW, "layer %d: event volume >= 2^24 cells or a dim beyond 32767", l);
        if (prev)
            for (int d = 0; d < s.geo.ndim; ++d)
                WFS_REQUIRE(s.geo.spatial[d] == prev->out_shape[d], WFS_EINVAL,
                            "layer %d: input shape is not layer %d's output shape", l, l - 1);
        prev = &s.geo;
        L.subm = s.geo.subm;
        L.in_direct = L.g.in_volume <= CH_GRID;
        L.out_direct = L.g.out_volume <= CH_GRID;
        L.nbr_out = s.nbr_out;
        L.nbr_in = s.nbr_in;
        L.out_indices = s.out_indices;
        L.N_cap = s.N_cap;
        L.M_cap = s.M_cap;
        L.m_dev = (long long *)s.m_dev;
        L.overflow_dev = s.overflow_dev;
        L.cell_ticket = s.cell_ticket;
        L.cell_row = s.cell_row;
    }
    return WFS_OK;
}

bool g_attr[2] = {false, false};

}  // namespace

extern "C" size_t wfs_rulebook_chain_workspace_bytes(int32_t batch_size) {
    return (size_t)256 + (size_t)(batch_size > 0 ? batch_size : 0) * CW * sizeof(int32_t);
}

extern "C" int wfs_rulebook_chain_count(const wfs_chain_layer *layers, int32_t nlayers, const int32_t *indices,
                                        int64_t N, const int64_t *n_dev, void *workspace, size_t workspace_bytes,
                                        int64_t *host_counts, int32_t *host_flags, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    Chain c;
    int rc = fill_chain(layers, nlayers, indices, N, n_dev, workspace, workspace_bytes, &c);
    if (rc != WFS_OK) return rc;
    WFS_REQUIRE(indices || N == 0, WFS_EINVAL, "NULL indices");
    WfsTimerScope timer(WFS_TIMER_RULEBOOK, stream);
    WFS_HIP_CHECK(hipMemsetAsync(c.flags, 0, 256, stream));
    if (!g_attr[0]) {
        WFS_HIP_CHECK(hipFuncSetAttribute((const void *)k_chain<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)CH_LDS_BYTES));
        g_attr[0] = true;
    }
    k_chain<true><<<dim3((unsigned)c.batch), dim3(CH_THREADS), CH_LDS_BYTES, stream>>>(c);
    WFS_LAUNCH_CHECK();
    if (!host_counts && !host_flags) return WFS_OK;
    // exact-size callers: read the per-layer totals back (synchronises)
    const size_t nb = (size_t)c.batch * CW;
    int *h = (int *)malloc(nb * sizeof(int));
    int hf[4] = {0, 0, 0, 0};
    WFS_REQUIRE(h, WFS_EINVAL, "out of host memory");
    hipError_t e1 = hipMemcpyAsync(h, c.counts, nb * sizeof(int), hipMemcpyDeviceToHost, stream);
    hipError_t e2 = hipMemcpyAsync(hf, c.flags, sizeof(hf), hipMemcpyDeviceToHost, stream);
    hipError_t e3 = hipStreamSynchronize(stream);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        free(h);
        wfs_set_error("reading the chain counts back failed");
        return WFS_EHIP;
    }
    long long nsum = 0;
    for (int l = 0; l < nlayers && host_counts; ++l) host_counts[l] = 0;
    for (int b = 0; b < c.batch; ++b) {
        nsum += h[(size_t)b * CW + 1];
        for (int l = 0; l < nlayers && host_counts; ++l) host_counts[l] += h[(size_t)b * CW + 2 + l];
    }
    free(h);
    if (nsum != N && !n_dev) hf[0] |= 1;
    if (host_flags) *host_flags = hf[0];
    return WFS_OK;
}

extern "C" int wfs_rulebook_chain_build(const wfs_chain_layer *layers, int32_t nlayers, const int32_t *indices,
                                        int64_t N, const int64_t *n_dev, void *workspace, size_t workspace_bytes,
                                        void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    Chain c;
    int rc = fill_chain(layers, nlayers, indices, N, n_dev, workspace, workspace_bytes, &c);
    if (rc != WFS_OK) return rc;
    for (int l = 0; l < nlayers; ++l) {
        const Layer &L = c.L[l];
        WFS_REQUIRE(L.nbr_out && L.N_cap >= 0, WFS_EINVAL, "layer %d: nbr_out is NULL", l);
        if (!L.subm) {
            WFS_REQUIRE(L.M_cap >= 0 && (L.M_cap == 0 || (L.nbr_in && L.out_indices)), WFS_EINVAL,
                        "layer %d: a regular conv needs nbr_in and out_indices", l);
            WFS_REQUIRE((L.cell_row == nullptr) == (L.cell_ticket == nullptr), WFS_EINVAL,
                        "layer %d: cell_row and cell_ticket come together", l);
        }
    }
    WfsTimerScope timer(WFS_TIMER_RULEBOOK, stream);
    if (!g_attr[1]) {
        WFS_HIP_CHECK(hipFuncSetAttribute((const void *)k_chain<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)CH_LDS_BYTES));
        g_attr[1] = true;
    }
    k_chain<false><<<dim3((unsigned)c.batch), dim3(CH_THREADS), CH_LDS_BYTES, stream>>>(c);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}
