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
constexpr int CH_SITES = 2048;            // rows of one event in any site set
constexpr int CH_HASH = 4096;             // hash slots per table (load <= 0.5)
constexpr int CH_GRID = 7168;             // an event's volume up to this uses a direct grid instead
constexpr int CH_TAB_WORDS = 2 * CH_GRID; // 56 KiB per table: direct [ticket | id] x cells, hash [key | ticket | id] x CH_HASH
constexpr unsigned EMPTY = 0xFFFFFFFFu;
static_assert(3 * CH_HASH <= CH_TAB_WORDS, "hash layout must fit the table memory");
constexpr size_t CH_LDS_BYTES = (size_t)(2 * CH_TAB_WORDS + 2 * CH_SITES) * 4;

struct Geo {
    int ndim, K;
    int spatial[4], out_shape[4], ksize[4], stride[4], padding[4], dilation[4];
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
    int *flags;                           // [0] error bits (1: rows not grouped by event / bad index, 2: event too large)
};
constexpr int CW = 2 + WFS_CHAIN_MAX_LAYERS;

// ---- site table in LDS -------------------------------------------------------------------------------------------
struct Tab {
    int *w;         // CH_TAB_WORDS words
    int direct;
    int vol;
    __device__ __forceinline__ int *keys() const { return w; }                                   // hash only
    __device__ __forceinline__ unsigned *tk() const { return (unsigned *)(w + (direct ? 0 : CH_HASH)); }
    __device__ __forceinline__ int *id() const { return w + (direct ? vol : 2 * CH_HASH); }
    __device__ __forceinline__ int slots() const { return direct ? vol : CH_HASH; }
    __device__ void clear() const {
        const int n = slots();
        unsigned *t = tk();
        int *d = id();
        for (int i = threadIdx.x; i < n; i += CH_THREADS) {
            t[i] = EMPTY;
            d[i] = -1;
            if (!direct) w[i] = -1;
        }
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
};
static_assert(CH_HASH == 4096, "hash shift above assumes 4096 slots");

// event-local keys: row-major over the layer's input / output shape, no batch term
__device__ __forceinline__ void decode(const int *shape, int ndim, int key, int *x) {
#pragma unroll
    for (int d = 3; d >= 0; --d) {
        if (d >= ndim) {
            x[d] = 0;
            continue;
        }
        x[d] = key % shape[d];
        key /= shape[d];
    }
}
__device__ __forceinline__ void offsets_of(const Geo &g, int k, int *off) {
#pragma unroll
    for (int d = 3; d >= 0; --d) {
        if (d >= g.ndim) {
            off[d] = 0;
            continue;
        }
        off[d] = k % g.ksize[d];
        k /= g.ksize[d];
    }
}
// output-site key of the candidate (input position x, offset k), or -1 (A.3 getValidOutPos: x + p - off*d = o*s)
__device__ __forceinline__ int out_key(const Geo &g, int k, const int *x) {
    int off[4];
    offsets_of(g, k, off);
    int lin = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (d >= g.ndim) break;
        int t = x[d] + g.padding[d] - off[d] * g.dilation[d];
        if (t < 0) return -1;
        int o = t / g.stride[d];
        if (o * g.stride[d] != t || o >= g.out_shape[d]) return -1;
        lin = lin * g.out_shape[d] + o;
    }
    return lin;
}
// input-site key that reaches output position o through offset k, or -1
__device__ __forceinline__ int in_key_of(const Geo &g, int k, const int *o) {
    int off[4];
    offsets_of(g, k, off);
    int lin = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (d >= g.ndim) break;
        int x = o[d] * g.stride[d] - g.padding[d] + off[d] * g.dilation[d];
        if (x < 0 || x >= g.spatial[d]) return -1;
        lin = lin * g.spatial[d] + x;
    }
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

template <bool COUNT>
__global__ void __launch_bounds__(CH_THREADS) k_chain(Chain c) {
    extern __shared__ __attribute__((aligned(16))) int lds[];
    __shared__ int sMisc[8];
    __shared__ int sW[CH_THREADS / 64];
    __shared__ long long sRed[2 * WFS_CHAIN_MAX_LAYERS + 2];
    int *keyA = lds, *keyB = lds + CH_SITES;
    Tab tin = {lds + 2 * CH_SITES, 0, 0}, tout = {lds + 2 * CH_SITES + CH_TAB_WORDS, 0, 0};
    const int e = blockIdx.x;
    const int tid = threadIdx.x;
    const int nd0 = c.L[0].g.ndim, stride = nd0 + 1;
    const long long nv = valid_rows(c.N, c.n_dev);
    int *cnt = c.counts + (long long)e * CW;

    long long start;
    int n;
    long long base[WFS_CHAIN_MAX_LAYERS];
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
#pragma unroll
        for (int l = 0; l < WFS_CHAIN_MAX_LAYERS; ++l) {
            // wave-level sums first, one LDS atomic per wave and quantity
            long long a = pre[l], t = tot[l];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                a += __shfl_xor(a, d, 64);
                t += __shfl_xor(t, d, 64);
            }
            if ((tid & 63) == 0) {
                atomicAdd((unsigned long long *)&sRed[2 * l], (unsigned long long)a);
                atomicAdd((unsigned long long *)&sRed[2 * l + 1], (unsigned long long)t);
            }
        }
        {
            long long s = nsum;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
            if ((tid & 63) == 0) atomicAdd((unsigned long long *)&sRed[2 * WFS_CHAIN_MAX_LAYERS], (unsigned long long)s);
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

    // ---- the event's input rows as keys over the first layer's input shape
    {
        const Geo &g = c.L[0].g;
        bool bad = false;
        for (int r = tid; r < n; r += CH_THREADS) {
            const int *row = c.idx + (start + r) * stride;
            bad = bad || row[0] != e;
            int lin = 0;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if (d >= g.ndim) break;
                const int x = row[1 + d];
                bad = bad || x < 0 || x >= g.spatial[d];
                lin = lin * g.spatial[d] + (x < 0 ? 0 : (x >= g.spatial[d] ? g.spatial[d] - 1 : x));
            }
            keyA[r] = lin;
        }
        if (COUNT && bad) atomicOr(&c.flags[0], 1);
    }
    long long in_base = start;
    bool in_built = false;
    tin.direct = c.L[0].in_direct;
    tin.vol = c.L[0].g.in_volume;
    __syncthreads();

    for (int l = 0; l < c.nlayers; ++l) {
        const Layer &L = c.L[l];
        const Geo &g = L.g;
        const int K = g.K;
        if (!COUNT && !in_built) {
            // input site table: key -> row of the event (duplicates: the last row wins, A.3)
            tin.clear();
            __syncthreads();
            for (int r = tid; r < n; r += CH_THREADS) atomicMax(&tin.id()[tin.insert(keyA[r])], r);
            __syncthreads();
            in_built = true;
        }
        if (L.subm) {
            if (!COUNT) {
                const int items = n * K;
                for (int i = tid; i < items; i += CH_THREADS) {
                    const int k = i / n, r = i - k * n;
                    int x[4];
                    decode(g.spatial, g.ndim, keyA[r], x);
                    const int nk = out_key(g, k, x);
                    int res = -1;
                    if (nk >= 0) {
                        const int s = tin.find(nk);
                        if (s >= 0) {
                            const int rr = tin.id()[s];
                            if (rr >= 0) res = (int)(in_base + rr);
                        }
                    }
                    L.nbr_out[(long long)k * L.N_cap + in_base + r] = res;
                }
            }
            continue;          // the site set is unchanged
        }
        // ---- regular conv: output site table
        tout.direct = L.out_direct;
        tout.vol = g.out_volume;
        tout.clear();
        if (tid == 0) sMisc[1] = 0;
        __syncthreads();
        {
            const int items = n * K;
            for (int i = tid; i < items; i += CH_THREADS) {
                const int k = i / n, r = i - k * n;
                int x[4];
                decode(g.spatial, g.ndim, keyA[r], x);
                const int ok = out_key(g, k, x);
                if (ok < 0) continue;
                const int s = tout.insert(ok);
                const unsigned old = atomicMin(&tout.tk()[s], (unsigned)(r * K + k));
                if (COUNT && old == EMPTY) {                 // a new site: append it (any order) for the next layer
                    const int pos = atomicAdd(&sMisc[1], 1);
                    if (pos < CH_SITES) keyB[pos] = ok;
                }
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
            // first-seen numbering: rows in order, offsets in order (A.3) = exclusive scan of first-ticket counts
            int carry = 0;
            for (int r0 = 0; r0 < n; r0 += CH_THREADS) {
                const int r = r0 + tid;
                unsigned mask = 0;
                int x[4];
                if (r < n) {
                    decode(g.spatial, g.ndim, keyA[r], x);
                    for (int k = 0; k < K; ++k) {
                        const int ok = out_key(g, k, x);
                        if (ok < 0) continue;
                        const int s = tout.find(ok);
                        if (s >= 0 && tout.tk()[s] == (unsigned)(r * K + k)) mask |= 1u << k;
                    }
                }
                int tot;
                int ex = block_excl_scan(__popc(mask), &tot, sW) + carry;
                while (mask) {
                    const int k = __builtin_ctz(mask);
                    mask &= mask - 1;
                    const int ok = out_key(g, k, x);
                    tout.id()[tout.find(ok)] = ex;
                    if (ex < CH_SITES) keyB[ex] = ok;
                    ++ex;
                }
                carry += tot;
                __syncthreads();
            }
            m = carry < CH_SITES ? carry : CH_SITES;          // carry > CH_SITES was flagged by the count kernel
            const long long out_base = base[l];
            // nbr_out: output row of (input row, offset)
            {
                const int items = n * K;
                for (int i = tid; i < items; i += CH_THREADS) {
                    const int k = i / n, r = i - k * n;
                    int x[4];
                    decode(g.spatial, g.ndim, keyA[r], x);
                    const int ok = out_key(g, k, x);
                    int res = -1;
                    if (ok >= 0) {
                        const int s = tout.find(ok);
                        const long long gid = out_base + tout.id()[s];
                        if (gid < L.M_cap) res = (int)gid;
                    }
                    L.nbr_out[(long long)k * L.N_cap + in_base + r] = res;
                }
            }
            // nbr_in: input row of (output row, offset), by looking the input position up; out_indices
            {
                const int items = m * K;
                for (int i = tid; i < items; i += CH_THREADS) {
                    const int k = i / m, o = i - k * m;
                    const long long gid = out_base + o;
                    if (gid >= L.M_cap) continue;
                    int ox[4];
                    decode(g.out_shape, g.ndim, keyB[o], ox);
                    const int ik = in_key_of(g, k, ox);
                    int res = -1;
                    if (ik >= 0) {
                        const int s = tin.find(ik);
                        if (s >= 0) {
                            const int rr = tin.id()[s];
                            if (rr >= 0) res = (int)(in_base + rr);
                        }
                    }
                    L.nbr_in[(long long)k * L.M_cap + gid] = res;
                }
                for (int o = tid; o < m; o += CH_THREADS) {
                    const long long gid = out_base + o;
                    if (gid >= L.M_cap) continue;
                    int ox[4];
                    decode(g.out_shape, g.ndim, keyB[o], ox);
                    int *dst = L.out_indices + gid * (g.ndim + 1);
                    dst[0] = e;
                    for (int d = 0; d < g.ndim; ++d) dst[1 + d] = ox[d];
                }
            }
            if (L.cell_row) {
                for (int cell = tid; cell < g.out_volume; cell += CH_THREADS) {
                    const int s = tout.find(cell);
                    const int o = s >= 0 ? tout.id()[s] : -1;
                    const long long gid = o >= 0 ? out_base + o : -1;
                    const bool ok = gid >= 0 && gid < L.M_cap;
                    L.cell_ticket[(long long)e * g.out_volume + cell] = ok ? 0u : EMPTY;
                    L.cell_row[(long long)e * g.out_volume + cell] = ok ? (int)gid : -1;
                }
            }
            in_base = out_base;
        }
        __syncthreads();
        // the outputs are the next layer's inputs; the output table (key -> local id) is its input table
        int *t = keyA;
        keyA = keyB;
        keyB = t;
        n = m;
        Tab tt = tin;
        tin = tout;
        tout = tt;
        in_built = !COUNT;
    }
}

bool geo_from(const wfs_geometry *g, Geo *G) {
    G->ndim = g->ndim;
    G->K = g->K;
    long long iv = 1, ov = 1;
    for (int i = 0; i < 4; ++i) {
        G->spatial[i] = g->spatial[i];
        G->out_shape[i] = g->out_shape[i];
        G->ksize[i] = g->ksize[i];
        G->stride[i] = g->stride[i];
        G->padding[i] = g->padding[i];
        G->dilation[i] = g->dilation[i];
        if (i < g->ndim) {
            iv *= g->spatial[i];
            ov *= g->out_shape[i];
        }
    }
    if (iv >= (1ll << 31) || ov >= (1ll << 31)) return false;
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
    c->counts = (int *)workspace + 64;
    const wfs_geometry *prev = nullptr;
    for (int l = 0; l < nlayers; ++l) {
        const wfs_chain_layer &s = layers[l];
        Layer &L = c->L[l];
        WFS_REQUIRE(s.geo.K >= 1 && s.geo.K <= 32, WFS_EINVAL, "layer %d: 1 <= K <= 32 (got %d)", l, s.geo.K);
        WFS_REQUIRE(s.geo.batch_size == batch && s.geo.ndim == layers[0].geo.ndim, WFS_EINVAL,
                    "layer %d: batch size / ndim differ from layer 0", l);
        WFS_REQUIRE(geo_from(&s.geo, &L.g), WFS_EOVERFLOW, "layer %d: event volume >= 2^31", l);
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
