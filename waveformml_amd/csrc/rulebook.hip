// rulebook.hip -- rulebook ("indice pairs") construction on gfx950.
//
// Replaces torch.ops.spconv.get_indice_pairs of spconv 1.2.1 (reference requirements.txt:15;
// call sites src/models/SPConvBlocks.py:75,134,498,803-810).  The CPU algorithm being matched
// bit-for-bit is sequential; the GPU formulation is order-free by construction:
//
//   * the candidate enumeration of getValidOutPos (out position descending per dim, last dim
//     fastest) visits kernel offsets in INCREASING linear offset k, so "candidate order" ==
//     "k order" and a candidate is the pair (input row j, offset k);
//   * SubM: hash[key] = j with "last wins" == atomicMax over j;
//   * regular conv: an output site's id is the rank of its FIRST ticket t = j*K + k among all
//     distinct sites == (number of first-tickets in rows < j) + (first-tickets of row j at
//     offsets < k): one atomicMin per candidate, one row-count, one exclusive scan over rows;
//   * spconv's indice_pairs[:, k, :n_k] is ordered by input row == stable compaction of the
//     gather-table column nbr_out[k, :] (wave ballot + popcount prefix, block offsets from a
//     scan over row tiles).
//
// Site lookup uses either a direct grid (slot == key, when batch*volume is small next to N) or an
// open-addressing hash of 32-bit keys (batch*volume < 2^31 is asserted by the front door).
#include <stdlib.h>

#include "wfs_common.h"

namespace {

constexpr int TB = 256;  // threads per block everywhere in this file

struct Geo {
    int ndim, K;
    int transposed;      // != 0: offsets are handled REVERSED inside the build (see offset_key), the tables flipped at the end
    int spatial[4], out_shape[4], ksize[4], stride[4], padding[4], dilation[4];
    long long in_volume, out_volume;
};

struct Table {       // site table: hash (direct == 0) or dense grid (direct == 1)
    int *keys;       // [cap] hash mode only, -1 = empty
    unsigned shift;  // 32 - log2(cap)
    unsigned mask;   // cap - 1
    int direct;
};

__device__ __forceinline__ unsigned tbl_home(const Table &t, int key) {
    return ((unsigned)key * 0x9E3779B1u) >> t.shift;
}
// find-or-insert; returns the slot
__device__ __forceinline__ unsigned tbl_insert(const Table &t, int key) {
    if (t.direct) return (unsigned)key;
    unsigned s = tbl_home(t, key);
    while (true) {
        int prev = atomicCAS(&t.keys[s], -1, key);
        if (prev == -1 || prev == key) return s;
        s = (s + 1) & t.mask;
    }
}
// find; returns slot or 0xFFFFFFFF
__device__ __forceinline__ unsigned tbl_find(const Table &t, int key) {
    if (t.direct) return (unsigned)key;
    unsigned s = tbl_home(t, key);
    while (true) {
        int cur = t.keys[s];
        if (cur == key) return s;
        if (cur == -1) return 0xFFFFFFFFu;
        s = (s + 1) & t.mask;
    }
}

// Per-row candidate walker: offsets k = 0..K-1 in increasing order (last dim fastest).
struct Walker {
    int off[4];
    __device__ __forceinline__ void reset() { off[0] = off[1] = off[2] = off[3] = 0; }
    __device__ __forceinline__ void next(const Geo &g) {
#pragma unroll
        for (int d = 3; d >= 0; --d) {
            if (d >= g.ndim) continue;
            if (++off[d] < g.ksize[d]) return;
            off[d] = 0;
        }
    }
    // output-site key of (row position x, batch b) at the current offset, or -1
    __device__ __forceinline__ int key(const Geo &g, const int *x, int b) const {
        long long lin = b;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if (d >= g.ndim) break;
            int o;
            if (g.transposed) {
                o = x[d] * g.stride[d] - g.padding[d] + (g.ksize[d] - 1 - off[d]) * g.dilation[d];
                if (o < 0 || o >= g.out_shape[d]) return -1;
            } else {
                int t = x[d] + g.padding[d] - off[d] * g.dilation[d];
                if (t < 0) return -1;
                o = t / g.stride[d];
                if (o * g.stride[d] != t || o >= g.out_shape[d]) return -1;
            }
            lin = lin * g.out_shape[d] + o;
        }
        return (int)lin;
    }
};

__device__ __forceinline__ bool load_row(const Geo &g, const int *idx, long long j, int *x, int &b, int batch) {
    const int *row = idx + j * (g.ndim + 1);
    b = row[0];
    bool ok = b >= 0 && b < batch;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (d < g.ndim) {
            x[d] = row[1 + d];
            ok = ok && x[d] >= 0 && x[d] < g.spatial[d];
        } else {
            x[d] = 0;
        }
    }
    return ok;
}

// key of an INPUT site (row-major over the input spatial shape)
__device__ __forceinline__ int in_key(const Geo &g, const int *x, int b) {
    long long lin = b;
#pragma unroll
    for (int d = 0; d < 4; ++d)
        if (d < g.ndim) lin = lin * g.spatial[d] + x[d];
    return (int)lin;
}

// ---------------------------------------------------------------- SubM
// info[0] = M, info[1] = duplicate coordinates seen, info[2] = out-of-range index seen
// Row counts: N is the CAPACITY (array strides, grid size); when n_dev is given the number of valid rows is
// read from device memory, so that a build never needs the host to know it (HIP-graph capturable steps).
__device__ __forceinline__ long long valid_rows(long long N, const long long *n_dev) {
    long long v = n_dev ? *n_dev : N;
    return v < N ? v : N;
}

__global__ void k_site_insert(Geo g, int batch, const int *idx, long long N, const long long *n_dev, Table t, int *vals,
                              long long *info) {
    long long j = (long long)blockIdx.x * TB + threadIdx.x;
    if (j >= valid_rows(N, n_dev)) return;
    int x[4], b;
    if (!load_row(g, idx, j, x, b, batch)) {
        info[2] = 1;
        return;
    }
    unsigned s = tbl_insert(t, in_key(g, x, b));
    int old = atomicMax(&vals[s], (int)j);      // duplicates: the LAST row wins (A.3)
    if (old >= 0) info[1] = 1;
}

// ---------------------------------------------------------------- (row, offset)-parallel forms
// grid = (rows / 64, ceil(K / 4)), block = 256: lane = row, wave = offset -> every thread does ONE site lookup /
// insert, writes of nbr_out[k][j] are coalesced over j, and the per-offset arithmetic is wave-uniform (scalar).
// K <= 32 lets a row's "first ticket" flags live in one 32-bit mask.
__device__ __forceinline__ int offset_key(const Geo &g, int k, const int *x, int b) {
    int rem = k;
    int off[4] = {0, 0, 0, 0};
#pragma unroll
    for (int d = 3; d >= 0; --d) {
        if (d >= g.ndim) continue;
        off[d] = rem % g.ksize[d];
        rem /= g.ksize[d];
    }
    long long lin = b;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (d >= g.ndim) break;
        int o;
        if (g.transposed) {
            // transposed conv: (x, offset) reaches x s - p + offset d.  spconv visits a row's offsets from the last to the
            // first; inside the build the offset index is REVERSED (k' = K - 1 - k) so that "first seen" keeps meaning
            // "smallest ticket j K + k'"; wfs_rulebook_emit flips the finished tables back to spconv's offset order
            o = x[d] * g.stride[d] - g.padding[d] + (g.ksize[d] - 1 - off[d]) * g.dilation[d];
            if (o < 0 || o >= g.out_shape[d]) return -1;
        } else {
            int t = x[d] + g.padding[d] - off[d] * g.dilation[d];
            if (t < 0) return -1;
            o = t / g.stride[d];
            if (o * g.stride[d] != t || o >= g.out_shape[d]) return -1;
        }
        lin = lin * g.out_shape[d] + o;
    }
    return (int)lin;
}

__global__ void __launch_bounds__(TB) k_subm_lookup2(Geo g, int batch, const int *__restrict__ idx, long long N,
                                                     const long long *n_dev, Table t, const int *__restrict__ vals,
                                                     int *__restrict__ nbr_out) {
    const int k = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
    const long long j = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    if (k >= g.K || j >= valid_rows(N, n_dev)) return;
    int x[4], b;
    bool ok = load_row(g, idx, j, x, b, batch);
    int res = -1;
    int key = ok ? offset_key(g, k, x, b) : -1;
    if (key >= 0) {
        unsigned s = tbl_find(t, key);
        if (s != 0xFFFFFFFFu) res = vals[s];
    }
    nbr_out[(long long)k * N + j] = res;
}

// Direct-grid SubM lookups by RUNS along the last dimension: the cells an input row looks up for the kl offsets of the
// last dimension (same offsets in the leading dimensions, dilation 1) are ADJACENT in the grid (key, key - 1, ...), so one
// thread serves a (row, leading offsets) pair: one row read and one key computation for kl neighbouring lookups instead
// of kl of each (the waveform nets' 3 x 3 x 3 kernels: 9 threads per row instead of 27).  Table rows stay coalesced over
// the rows.  q = leading-offset index (offset k = q * kl + o, last dimension fastest, as everywhere).
constexpr int RUN_MAX = 8;
__global__ void __launch_bounds__(TB) k_subm_lookup_runs(Geo g, int batch, const int *__restrict__ idx, long long N,
                                                         const long long *n_dev, const int *__restrict__ vals,
                                                         int *__restrict__ nbr_out) {
    const int last = g.ndim - 1;
    const int kl = g.ksize[last];
    const int q = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
    const long long j = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    if (q * kl >= g.K || j >= valid_rows(N, n_dev)) return;
    int x[4], b;
    bool ok = load_row(g, idx, j, x, b, batch);
    // leading offsets of q (row-major over the leading kernel dims), prefix of the key
    int rem = q;
    long long lin = b;
    int off[4] = {0, 0, 0, 0};
#pragma unroll
    for (int d = 3; d >= 0; --d) {
        if (d >= last) continue;
        off[d] = rem % g.ksize[d];
        rem /= g.ksize[d];
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (d >= last) break;
        const int o = x[d] + g.padding[d] - off[d] * g.dilation[d];
        ok = ok && o >= 0 && o < g.out_shape[d];
        lin = lin * g.out_shape[d] + o;
    }
    const int xl = x[last] + g.padding[last];                 // cell of offset 0 in the last dimension; offset o: xl - o
    const long long base = lin * g.out_shape[last];
    int res[RUN_MAX];
#pragma unroll
    for (int o = 0; o < RUN_MAX; ++o) {
        const int c = xl - o;
        const bool in = ok && o < kl && c >= 0 && c < g.out_shape[last];
        res[o] = in ? vals[base + c] : -1;
    }
#pragma unroll
    for (int o = 0; o < RUN_MAX; ++o)
        if (o < kl) nbr_out[(long long)(q * kl + o) * N + j] = res[o];
}

// the (row, offset)-parallel kernels keep 32-bit tickets (the host checks N*K < 2^32): native 32-bit atomicMin
__global__ void __launch_bounds__(TB) k_conv_insert2(Geo g, int batch, const int *__restrict__ idx, long long N,
                                                     const long long *n_dev, Table t, unsigned *__restrict__ ticket,
                                                     int *__restrict__ nbr_out, long long *info) {
    const int k = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
    const long long j = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    if (k >= g.K || j >= valid_rows(N, n_dev)) return;
    int x[4], b;
    bool ok = load_row(g, idx, j, x, b, batch);
    if (!ok) info[2] = 1;
    int key = ok ? offset_key(g, k, x, b) : -1;
    int slot = -1;
    if (key >= 0) {
        unsigned s = tbl_insert(t, key);
        atomicMin(&ticket[s], (unsigned)(j * g.K + k));
        slot = (int)s;
    }
    nbr_out[(long long)k * N + j] = slot;
}

// The same insert with one thread per (row, leading offsets): along the LAST dimension (dilation 1, stride s) only the
// offsets o = (x + p) mod s, + s, ... reach an output cell at all -- for the waveform nets' k = 3, s = 4 layers at most
// one of the three, and none for a quarter of the rows -- so a thread visits just those and writes -1 for the rest,
// instead of three threads each loading the row and dividing to find out.  Same tickets, same table.
__global__ void __launch_bounds__(TB) k_conv_insert_runs(Geo g, int batch, const int *__restrict__ idx, long long N,
                                                         const long long *n_dev, Table t, unsigned *__restrict__ ticket,
                                                         int *__restrict__ nbr_out, long long *info) {
    const int last = g.ndim - 1;
    const int kl = g.ksize[last], sl = g.stride[last];
    const int q = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
    const long long j = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    if (q * kl >= g.K || j >= valid_rows(N, n_dev)) return;
    int x[4], b;
    bool ok = load_row(g, idx, j, x, b, batch);
    if (!ok) info[2] = 1;
    int rem = q;
    long long lin = b;
    int off[4] = {0, 0, 0, 0};
#pragma unroll
    for (int d = 3; d >= 0; --d) {
        if (d >= last) continue;
        off[d] = rem % g.ksize[d];
        rem /= g.ksize[d];
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (d >= last) break;
        const int tt = x[d] + g.padding[d] - off[d] * g.dilation[d];
        const int o = tt >= 0 ? tt / g.stride[d] : -1;
        ok = ok && tt >= 0 && o * g.stride[d] == tt && o < g.out_shape[d];
        lin = lin * g.out_shape[d] + (o >= 0 ? o : 0);
    }
    const int xl = x[last] + g.padding[last];
    const int r0 = xl % sl;                          // the offsets that divide: r0, r0 + sl, ...
    const long long base = lin * g.out_shape[last];
#pragma unroll
    for (int o = 0; o < RUN_MAX; ++o) {
        if (o >= kl) break;
        int slot = -1;
        const int tt = xl - o;
        if (ok && o >= r0 && (o - r0) % sl == 0 && tt >= 0) {
            const int oc = tt / sl;
            if (oc < g.out_shape[last]) {
                const int k = q * kl + o;
                const unsigned sidx = tbl_insert(t, (int)(base + oc));
                atomicMin(&ticket[sidx], (unsigned)(j * g.K + k));
                slot = (int)sidx;
            }
        }
        nbr_out[(long long)(q * kl + o) * N + j] = slot;
    }
}

__global__ void __launch_bounds__(TB) k_conv_assign2(Geo g, long long N, const long long *n_dev, long long M_cap,
                                                     const int *__restrict__ nbr_out, const unsigned *__restrict__ rowmask,
                                                     const int *__restrict__ rowbase, Table t, int *__restrict__ slot_id,
                                                     int *__restrict__ out_indices, long long *info,
                                                     int *__restrict__ nbr_in_fill, long long n_fill) {
    // nbr_in = -1 everywhere, for the atomicMax of k_conv_finalize2 (the NEXT launch): saves a memset launch
    if (nbr_in_fill) {
        const long long nthreads = (long long)gridDim.x * gridDim.y * TB;
        for (long long i = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * TB + threadIdx.x; i < n_fill; i += nthreads)
            nbr_in_fill[i] = -1;
    }
    const int k = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
    const long long j = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    if (k >= g.K || j >= valid_rows(N, n_dev)) return;
    const unsigned m = rowmask[j];
    if (!((m >> k) & 1u)) return;
    const int s = nbr_out[(long long)k * N + j];
    const int id = rowbase[j] + __popc(m & ((1u << k) - 1u));       // first-seen order: rows, then offsets
    slot_id[s] = id;
    if (id >= M_cap) {
        info[3] = 1;
        return;
    }
    long long key = t.direct ? (long long)s : (long long)t.keys[s];
    int *o = out_indices + (long long)id * (g.ndim + 1);
    for (int d = g.ndim - 1; d >= 0; --d) {
        o[1 + d] = (int)(key % g.out_shape[d]);
        key /= g.out_shape[d];
    }
    o[0] = (int)key;
}

__global__ void k_set_if(const long long *__restrict__ cond, int *__restrict__ flag) {
    if (*cond != 0) *flag = 1;
}

__global__ void __launch_bounds__(TB) k_conv_finalize2(int K, long long N, const long long *n_dev, long long M,
                                                       int *__restrict__ nbr_out, const int *__restrict__ slot_id,
                                                       int *__restrict__ nbr_in, const long long *__restrict__ info,
                                                       int *__restrict__ overflow) {
    const int k = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
    const long long j = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    // STICKY: set when M exceeded the capacity, never cleared here (the reader clears it)
    if (overflow && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && info[0] > M) *overflow = 1;
    if (k >= K || j >= valid_rows(N, n_dev)) return;
    int s = nbr_out[(long long)k * N + j];
    if (s < 0) return;
    int id = slot_id[s];
    if (id >= M) id = -1;
    nbr_out[(long long)k * N + j] = id;
    if (nbr_in && id >= 0) atomicMax(&nbr_in[(long long)k * M + id], (int)j);
}

// ---------------------------------------------------------------- regular / strided conv
__global__ void k_conv_insert(Geo g, int batch, const int *idx, long long N, const long long *n_dev, Table t,
                              unsigned long long *ticket, int *nbr_out, long long *info) {
    long long j = (long long)blockIdx.x * TB + threadIdx.x;
    if (j >= valid_rows(N, n_dev)) return;
    int x[4], b;
    bool ok = load_row(g, idx, j, x, b, batch);
    if (!ok) info[2] = 1;
    Walker w;
    w.reset();
    for (int k = 0; k < g.K; ++k) {
        int key = ok ? w.key(g, x, b) : -1;
        int slot = -1;
        if (key >= 0) {
            unsigned s = tbl_insert(t, key);
            atomicMin(&ticket[s], (unsigned long long)j * g.K + k);
            slot = (int)s;
        }
        nbr_out[(long long)k * N + j] = slot;   // slot for now; k_conv_finalize turns it into the id
        w.next(g);
    }
}

__global__ void k_conv_rowcount(int K, long long N, const long long *n_dev, const int *nbr_out,
                                const unsigned long long *ticket, int *rowfirst) {
    long long j = (long long)blockIdx.x * TB + threadIdx.x;
    if (j >= N) return;
    if (j >= valid_rows(N, n_dev)) {       // the scan runs over the capacity
        rowfirst[j] = 0;
        return;
    }
    int c = 0;
    for (int k = 0; k < K; ++k) {
        int s = nbr_out[(long long)k * N + j];
        if (s >= 0 && ticket[s] == (unsigned long long)j * K + k) ++c;
    }
    rowfirst[j] = c;
}

__global__ void k_conv_assign(Geo g, long long N, const long long *n_dev, long long M_cap, const int *nbr_out,
                              const unsigned long long *ticket, const int *rowbase, Table t, int *slot_id,
                              int *out_indices, long long *info) {
    long long j = (long long)blockIdx.x * TB + threadIdx.x;
    if (j >= valid_rows(N, n_dev)) return;
    int id = rowbase[j];
    for (int k = 0; k < g.K; ++k) {
        int s = nbr_out[(long long)k * N + j];
        if (s >= 0 && ticket[s] == (unsigned long long)j * g.K + k) {
            slot_id[s] = id;
            if (id >= M_cap) {                 // output capacity exceeded (device-count mode): flag, do not write
                info[3] = 1;
                ++id;
                continue;
            }
            long long key = t.direct ? (long long)s : (long long)t.keys[s];
            int *o = out_indices + (long long)id * (g.ndim + 1);
            for (int d = g.ndim - 1; d >= 0; --d) {
                o[1 + d] = (int)(key % g.out_shape[d]);
                key /= g.out_shape[d];
            }
            o[0] = (int)key;
            ++id;
        }
    }
}

__global__ void k_conv_finalize(int K, long long N, const long long *n_dev, long long M, int *nbr_out,
                                const int *slot_id, int *nbr_in) {
    long long j = (long long)blockIdx.x * TB + threadIdx.x;
    if (j >= valid_rows(N, n_dev)) return;
    for (int k = 0; k < K; ++k) {
        int s = nbr_out[(long long)k * N + j];
        if (s >= 0) {
            int id = slot_id[s];
            if (id >= M) id = -1;              // beyond the output capacity (flagged by k_conv_assign)
            nbr_out[(long long)k * N + j] = id;
            if (nbr_in && id >= 0) atomicMax(&nbr_in[(long long)k * M + id], (int)j);
        }
    }
}

// SubM with an even kernel or dilation: nbr_in by scatter of nbr_out
__global__ void k_invert_table(int K, long long N, const long long *n_dev, long long M, const int *nbr_out,
                               int *nbr_in) {
    long long j = (long long)blockIdx.x * TB + threadIdx.x;
    if (j >= valid_rows(N, n_dev)) return;
    for (int k = 0; k < K; ++k) {
        int i = nbr_out[(long long)k * N + j];
        if (i >= 0) atomicMax(&nbr_in[(long long)k * M + i], (int)j);
    }
}

// ---------------------------------------------------------------- exclusive scan (int32)
// three launches: per-block sums -> scan of block sums (one block) -> per-block exclusive scan
constexpr int SCAN_ITEMS = 4;                      // items per thread
constexpr int SCAN_TILE = TB * SCAN_ITEMS;

__device__ __forceinline__ int wave_incl_scan(int v) {
    int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}
// block-wide exclusive scan of one int per thread; returns exclusive prefix, *total = block sum
__device__ __forceinline__ int block_excl_scan(int v, int *total) {
    __shared__ int wsum[TB / 64];
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = wave_incl_scan(v);
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < TB / 64; ++w) {
        int s = wsum[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// popc != 0: the scanned quantity is the popcount of each input word (row masks of first tickets); rows at or
// beyond the valid count contribute 0
__device__ __forceinline__ int scan_item(const int *in, long long i, long long n, long long nv, int popc) {
    if (i >= n) return 0;
    if (!popc) return in[i];
    return i < nv ? __popc((unsigned)in[i]) : 0;
}

__global__ void k_scan_blocksum(const int *in, long long n, const long long *n_dev, int popc, int *bsum) {
    long long base = (long long)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    const long long nv = valid_rows(n, n_dev);
    int v = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) v += scan_item(in, base + i, n, nv, popc);
    int tot;
    block_excl_scan(v, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
// single block: exclusive scan of bsum[0..nb) in place; total -> *total (long long) and total32
__global__ void k_scan_top(int *bsum, long long nb, long long *total, long long *total2, long long cap) {
    int carry = 0;
    for (long long base = 0; base < nb; base += TB) {
        long long i = base + threadIdx.x;
        int v = i < nb ? bsum[i] : 0;
        int tot;
        int ex = block_excl_scan(v, &tot);
        if (i < nb) bsum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        *total = carry;
        if (total2) *total2 = carry < cap ? carry : cap;       // the row count downstream kernels bound by
    }
}
__global__ void k_scan_apply(const int *in, long long n, const long long *n_dev, int popc, const int *bsum, int *out) {
    long long base = (long long)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    const long long nv = valid_rows(n, n_dev);
    int vals[SCAN_ITEMS];
    int v = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        vals[i] = scan_item(in, base + i, n, nv, popc);
        v += vals[i];
    }
    int tot;
    int ex = block_excl_scan(v, &tot) + bsum[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = ex;
        ex += vals[i];
    }
}

// ---------------------------------------------------------------- fewer, fatter launches for the K <= 32 path
// At the PSD batch sizes a launch costs ~5 us whatever it does, and the strided layers' rulebook chain sits on the
// critical path of the step (their builds run beside the first layers on a side stream and must be done when those
// are): ten launches per rulebook became six.

// workspace initialisation in one launch: info[0..3] = 0, `ones` region (hash keys, values / tickets) = 0xFF
__global__ void __launch_bounds__(TB) k_ws_init(long long *__restrict__ info, uint4 *__restrict__ ones, long long n16) {
    if (blockIdx.x == 0 && threadIdx.x < 4) info[threadIdx.x] = 0;
    const uint4 v = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (long long i = (long long)blockIdx.x * TB + threadIdx.x; i < n16; i += (long long)gridDim.x * TB) ones[i] = v;
}

// row-parallel: rowmask[j] bit k = candidate (j, k) holds the first ticket of its site; bsum[block] = number of
// first tickets of the block's TB rows (= output sites they introduce).  No atomics, rowmask needs no clearing.
__global__ void __launch_bounds__(TB) k_conv_first_bsum(int K, long long N, const long long *n_dev,
                                                        const int *__restrict__ nbr_out,
                                                        const unsigned *__restrict__ ticket,
                                                        unsigned *__restrict__ rowmask, int *__restrict__ bsum) {
    const long long j = (long long)blockIdx.x * TB + threadIdx.x;
    const long long nv = valid_rows(N, n_dev);
    const bool live = j < nv;
    const long long jc = live ? j : 0;
    int s[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) s[k] = nbr_out[(long long)(k < K ? k : K - 1) * N + jc];
    unsigned t[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) t[k] = ticket[s[k] >= 0 ? s[k] : 0];
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k)
        if (k < K && live && s[k] >= 0 && t[k] == (unsigned)(j * K + k)) m |= 1u << k;
    if (j < N) rowmask[j] = m;
    int tot;
    block_excl_scan(__popc(m), &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// rowbase[j] = number of output sites introduced by rows < j (exclusive scan of popc(rowmask)): every block adds up
// the block sums in front of it (<= a few hundred ints) instead of waiting for a separate top-level scan launch; the
// last block publishes the totals: *total = M, *total2 = min(M, cap) = the row count downstream kernels bound by.
__global__ void __launch_bounds__(TB) k_conv_rowbase(long long N, const unsigned *__restrict__ rowmask,
                                                     const int *__restrict__ bsum, int *__restrict__ rowbase,
                                                     long long *total, long long *total2, long long cap) {
    __shared__ int sPre;
    int acc = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += TB) acc += bsum[i];
    int pre;
    block_excl_scan(acc, &pre);
    if (threadIdx.x == 0) sPre = pre;
    __syncthreads();
    const long long j = (long long)blockIdx.x * TB + threadIdx.x;
    const int v = j < N ? __popc(rowmask[j]) : 0;           // rows beyond the valid count hold an empty mask
    int tot;
    const int ex = block_excl_scan(v, &tot);
    if (j < N) rowbase[j] = sPre + ex;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const long long M = (long long)sPre + tot;
        *total = M;
        if (total2) *total2 = M < cap ? M : cap;
    }
}

// rows k <-> K - 1 - k of a [K, n] table (transposed conv: the build works with reversed offsets)
__global__ void __launch_bounds__(TB) k_flip_rows(int K, long long n, const long long *n_dev, int *__restrict__ t) {
    const long long j = (long long)blockIdx.x * TB + threadIdx.x;
    const int k = blockIdx.y;
    const long long nv = n_dev ? (*n_dev < n ? *n_dev : n) : n;
    if (j >= nv || 2 * k + 1 >= K) return;
    const int a = t[(long long)k * n + j], b = t[(long long)(K - 1 - k) * n + j];
    t[(long long)k * n + j] = b;
    t[(long long)(K - 1 - k) * n + j] = a;
}

// ---------------------------------------------------------------- compaction to spconv's encoding
// tile = TB consecutive input rows.  tcount[k * ntiles + tile] = valid entries of column k in tile.
__global__ void k_compact_count(int K, long long N, const long long *n_dev, long long ntiles, const int *nbr_out,
                                int *tcount) {
    __shared__ int wsum[TB / 64];
    long long j = (long long)blockIdx.x * TB + threadIdx.x;
    const long long Nv = valid_rows(N, n_dev);
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int k = 0; k < K; ++k) {
        bool v = j < Nv && nbr_out[(long long)k * N + j] >= 0;
        unsigned long long m = __ballot(v);
        if (lane == 0) wsum[wid] = __popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            int s = 0;
#pragma unroll
            for (int w = 0; w < TB / 64; ++w) s += wsum[w];
            tcount[(long long)k * ntiles + blockIdx.x] = s;
        }
        __syncthreads();
    }
}
// one block per offset k: exclusive scan over tiles in place, total -> pair_num[k]
__global__ void k_compact_scan(long long ntiles, int *tcount, int *pair_num) {
    int *row = tcount + (long long)blockIdx.x * ntiles;
    int carry = 0;
    for (long long base = 0; base < ntiles; base += TB) {
        long long i = base + threadIdx.x;
        int v = i < ntiles ? row[i] : 0;
        int tot;
        int ex = block_excl_scan(v, &tot);
        if (i < ntiles) row[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) pair_num[blockIdx.x] = carry;
}
__global__ void k_compact_write(int K, long long N, const long long *n_dev, long long ntiles, const int *nbr_out,
                                const int *tcount, const int *pair_num, int *pairs) {
    __shared__ int wsum[TB / 64];
    long long j = (long long)blockIdx.x * TB + threadIdx.x;
    const long long Nv = valid_rows(N, n_dev);
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int k = 0; k < K; ++k) {
        int o = j < Nv ? nbr_out[(long long)k * N + j] : -1;
        bool v = o >= 0;
        unsigned long long m = __ballot(v);
        if (lane == 0) wsum[wid] = __popcll(m);
        __syncthreads();
        int base = tcount[(long long)k * ntiles + blockIdx.x];
#pragma unroll
        for (int w = 0; w < TB / 64; ++w)
            if (w < wid) base += wsum[w];
        int rank = __popcll(m & ((1ull << lane) - 1ull));
        int *p0 = pairs + (long long)k * N;
        int *p1 = pairs + ((long long)K + k) * N;
        if (v) {
            p0[base + rank] = (int)j;
            p1[base + rank] = o;
        }
        if (j < N && j >= pair_num[k]) {       // the -1 padding of A.2, written once
            p0[j] = -1;
            p1[j] = -1;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- host side
struct Plan {
    Geo geo;
    Table tbl;
    long long cap;
    // workspace carve-up (byte offsets)
    size_t off_info, off_keys, off_vals, off_ticket, off_slot_id, off_rowfirst, off_rowbase, off_bsum,
        off_tcount, total, zero_bytes, ones_bytes;
    long long ntiles, nscan;
};

unsigned log2_ceil(unsigned long long v) {
    unsigned l = 0;
    while ((1ull << l) < v) ++l;
    return l;
}

void make_plan(const wfs_geometry *g, long long N, Plan *p) {
    Geo &G = p->geo;
    G.ndim = g->ndim;
    G.K = g->K;
    G.transposed = g->transposed ? 1 : 0;
    G.in_volume = G.out_volume = 1;
    for (int i = 0; i < 4; ++i) {
        G.spatial[i] = g->spatial[i];
        G.out_shape[i] = g->out_shape[i];
        G.ksize[i] = g->ksize[i];
        G.stride[i] = g->stride[i];
        G.padding[i] = g->padding[i];
        G.dilation[i] = g->dilation[i];
        if (i < g->ndim) {
            G.in_volume *= g->spatial[i];
            G.out_volume *= g->out_shape[i];
        }
    }
    long long cells = (long long)g->batch_size * G.out_volume;   // < 2^31
    long long bound = g->subm ? N : (N * (long long)g->K < cells ? N * (long long)g->K : cells);
    if (bound < 1) bound = 1;
    // direct grid (one int per site of the batch) instead of a hash table: always when it is at most 4x the bound, and
    // for SubM also up to 16 M sites / 128x the bound -- measured at the PSD shape (10 M sites, 10^5 rows): clearing
    // the 40 MB grid costs 10 us, but the insert drops 13 -> 5 us and the 2.3 M lookups 24 -> 19 us (one read each, no
    // key compare, no probing); a 20 M-site grid would lose to the hash table again
    p->tbl.direct = cells <= 4 * bound || (g->subm && cells <= (1ll << 24) && cells <= 128 * bound);
    if (p->tbl.direct) {
        p->cap = cells > 0 ? cells : 1;
        p->tbl.shift = 0;
        p->tbl.mask = 0;
    } else {
        unsigned lg = log2_ceil((unsigned long long)(2 * bound));
        if (lg < 6) lg = 6;
        p->cap = 1ll << lg;
        p->tbl.shift = 32 - lg;
        p->tbl.mask = (unsigned)(p->cap - 1);
    }
    p->ntiles = wfs_cdiv(N > 0 ? N : 1, TB);
    p->nscan = wfs_cdiv(N > 0 ? N : 1, SCAN_TILE);
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t at = o;
        o = wfs_align_up(o + bytes, 256);
        return at;
    };
    // zero region: info, then (regular conv) the per-row first-ticket words
    p->off_info = take(4 * sizeof(long long));
    p->off_rowfirst = g->subm ? o : take((size_t)(N + 1) * 4);
    p->zero_bytes = o - p->off_info;
    // 0xFF region: hash keys, then SubM values or regular-conv tickets
    p->off_keys = take(p->tbl.direct ? 0 : (size_t)p->cap * 4);
    if (g->subm) {
        p->off_vals = take((size_t)p->cap * 4);
        p->ones_bytes = o - p->off_keys;
        p->off_ticket = p->off_slot_id = p->off_rowbase = p->off_bsum = o;
    } else {
        p->off_vals = o;
        p->off_ticket = take((size_t)p->cap * 8);
        p->ones_bytes = o - p->off_keys;
        p->off_slot_id = take((size_t)p->cap * 4);
        p->off_rowbase = take((size_t)(N + 1) * 4);
        p->off_bsum = take((size_t)p->ntiles * 4);          // one block sum per TB rows (>= nscan)
    }
    p->off_tcount = take((size_t)g->K * p->ntiles * 4);
    p->total = o;
}

}  // namespace

extern "C" size_t wfs_rulebook_workspace_bytes(const wfs_geometry *g, int64_t N) {
    if (!g || N < 0) return 0;
    Plan p;
    make_plan(g, N, &p);
    return p.total;
}

// Standalone duplicate check for index sets whose uniqueness is unknown before a regular conv
// (a regular conv's OUTPUT is unique by construction; SubM learns it for free in its plan).
// workspace: same size as a SubM rulebook over (spatial, batch).  Synchronises `stream`.
extern "C" int wfs_indices_check(const wfs_geometry *g_subm, const int32_t *indices, int64_t N,
                                 void *workspace, size_t workspace_bytes, int64_t host_info[2], void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(g_subm && g_subm->subm, WFS_EINVAL, "wfs_indices_check wants a SubM geometry");
    Plan p;
    make_plan(g_subm, N, &p);
    WFS_REQUIRE(workspace_bytes >= p.total, WFS_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, p.total);
    char *ws = (char *)workspace;
    long long *info = (long long *)(ws + p.off_info);
    p.tbl.keys = (int *)(ws + p.off_keys);
    int *vals = (int *)(ws + p.off_vals);
    WFS_HIP_CHECK(hipMemsetAsync(ws + p.off_info, 0, p.zero_bytes, stream));
    WFS_HIP_CHECK(hipMemsetAsync(ws + p.off_keys, 0xFF, p.ones_bytes, stream));
    if (N > 0) {
        k_site_insert<<<dim3((unsigned)wfs_cdiv(N, TB)), dim3(TB), 0, stream>>>(p.geo, g_subm->batch_size, indices, N,
                                                                              nullptr, p.tbl, vals, info);
        WFS_LAUNCH_CHECK();
    }
    long long h[4];
    WFS_HIP_CHECK(hipMemcpyAsync(h, info, sizeof(h), hipMemcpyDeviceToHost, stream));
    WFS_HIP_CHECK(hipStreamSynchronize(stream));
    WFS_REQUIRE(h[2] == 0, WFS_EINVAL, "an index row lies outside batch_size/spatial_shape");
    host_info[0] = N;
    host_info[1] = h[1];
    return WFS_OK;
}

extern "C" int wfs_rulebook_plan(const wfs_geometry *g, const int32_t *indices, int64_t N, int32_t *nbr_out,
                                 void *workspace, size_t workspace_bytes, int64_t host_info[2],
                                 const int64_t *n_dev, int64_t *m_dev, int64_t M_cap, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(g, WFS_EINVAL, "NULL argument");
    WFS_REQUIRE(host_info || g->subm || m_dev, WFS_EINVAL,
                "a regular conv must hand M back: give host_info (synchronises) or m_dev (device)");
    WFS_REQUIRE(N >= 0 && N < (1ll << 31), WFS_EINVAL, "N out of range");
    WFS_REQUIRE(g->K >= 1, WFS_EINVAL, "geometry not initialised (wfs_geometry_init)");
    if (host_info) {
        host_info[0] = g->subm ? N : 0;
        host_info[1] = 0;
    }
    if (N == 0) {
        if (m_dev) WFS_HIP_CHECK(hipMemsetAsync(m_dev, 0, sizeof(int64_t), stream));
        return WFS_OK;
    }
    WFS_REQUIRE(indices && nbr_out && workspace, WFS_EINVAL, "NULL device pointer");
    Plan p;
    make_plan(g, N, &p);
    WFS_REQUIRE(workspace_bytes >= p.total, WFS_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, p.total);
    WfsTimerScope timer(WFS_TIMER_RULEBOOK, stream);
    const long long *nd = (const long long *)n_dev;
    char *ws = (char *)workspace;
    long long *info = (long long *)(ws + p.off_info);
    p.tbl.keys = (int *)(ws + p.off_keys);
    dim3 grid((unsigned)wfs_cdiv(N, TB)), block(TB);
    const int wide = !g->subm && g->K <= 32 && (long long)N * g->K < (1ll << 32);
    if (g->subm || wide) {
        // info = 0 and keys / values / tickets = 0xFF in one launch (the row masks of the wide path need no clearing)
        const long long n16 = (long long)(p.ones_bytes / 16);
        long long ib = wfs_cdiv(n16 > 0 ? n16 : 1, TB * 4);
        if (ib > 2048) ib = 2048;
        k_ws_init<<<dim3((unsigned)ib), block, 0, stream>>>(info, (uint4 *)(ws + p.off_keys), n16);
        WFS_LAUNCH_CHECK();
    } else {
        WFS_HIP_CHECK(hipMemsetAsync(ws + p.off_info, 0, p.zero_bytes, stream));        // info + first-ticket words
        WFS_HIP_CHECK(hipMemsetAsync(ws + p.off_keys, 0xFF, p.ones_bytes, stream));     // keys + tickets
    }
    if (g->subm) {
        int *vals = (int *)(ws + p.off_vals);
        k_site_insert<<<grid, block, 0, stream>>>(p.geo, g->batch_size, indices, N, nd, p.tbl, vals, info);
        WFS_LAUNCH_CHECK();
        const int kl = g->ksize[g->ndim - 1];
        static const bool runs_on = [] { const char *e = getenv("WFS_SUBM_RUNS"); return !(e && e[0] == '0'); }();
        if (runs_on && p.tbl.direct && g->dilation[g->ndim - 1] == 1 && kl >= 2 && kl <= RUN_MAX) {
            dim3 grid_r((unsigned)wfs_cdiv(N, 64), (unsigned)wfs_cdiv(g->K / kl, 4));
            k_subm_lookup_runs<<<grid_r, block, 0, stream>>>(p.geo, g->batch_size, indices, N, nd, vals, nbr_out);
        } else {
            dim3 grid2((unsigned)wfs_cdiv(N, 64), (unsigned)wfs_cdiv(g->K, 4));
            k_subm_lookup2<<<grid2, block, 0, stream>>>(p.geo, g->batch_size, indices, N, nd, p.tbl, vals, nbr_out);
        }
        WFS_LAUNCH_CHECK();
        if (m_dev && n_dev && m_dev != n_dev)
            WFS_HIP_CHECK(hipMemcpyAsync(m_dev, n_dev, sizeof(int64_t), hipMemcpyDeviceToDevice, stream));
    } else {
        unsigned long long *ticket = (unsigned long long *)(ws + p.off_ticket);
        int *rowfirst = (int *)(ws + p.off_rowfirst);
        int *rowbase = (int *)(ws + p.off_rowbase);
        int *bsum = (int *)(ws + p.off_bsum);
        // (row, offset)-parallel insert; rowfirst[] then holds first-ticket masks, tickets are 32-bit
        dim3 grid2((unsigned)wfs_cdiv(N, 64), (unsigned)wfs_cdiv(g->K, 4));
        const long long mcap = M_cap > 0 ? M_cap : (1ll << 62);
        if (wide) {
            const int kl = g->ksize[g->ndim - 1];
            static const bool runs_on = [] { const char *e = getenv("WFS_CONV_RUNS"); return !(e && e[0] == '0'); }();
            if (runs_on && !g->transposed && g->dilation[g->ndim - 1] == 1 && kl >= 2 && kl <= RUN_MAX) {
                dim3 grid_r((unsigned)wfs_cdiv(N, 64), (unsigned)wfs_cdiv(g->K / kl, 4));
                k_conv_insert_runs<<<grid_r, block, 0, stream>>>(p.geo, g->batch_size, indices, N, nd, p.tbl,
                                                                 (unsigned *)ticket, nbr_out, info);
            } else {
                k_conv_insert2<<<grid2, block, 0, stream>>>(p.geo, g->batch_size, indices, N, nd, p.tbl, (unsigned *)ticket,
                                                            nbr_out, info);
            }
            WFS_LAUNCH_CHECK();
            k_conv_first_bsum<<<grid, block, 0, stream>>>(g->K, N, nd, nbr_out, (const unsigned *)ticket,
                                                          (unsigned *)rowfirst, bsum);
            WFS_LAUNCH_CHECK();
            // info[0] = M; m_dev = min(M, M_cap) = the row count every consumer of the outputs is bounded by
            k_conv_rowbase<<<grid, block, 0, stream>>>(N, (const unsigned *)rowfirst, bsum, rowbase, info,
                                                       (long long *)m_dev, mcap);
            WFS_LAUNCH_CHECK();
        } else {
            k_conv_insert<<<grid, block, 0, stream>>>(p.geo, g->batch_size, indices, N, nd, p.tbl, ticket, nbr_out, info);
            WFS_LAUNCH_CHECK();
            k_conv_rowcount<<<grid, block, 0, stream>>>(g->K, N, nd, nbr_out, ticket, rowfirst);
            WFS_LAUNCH_CHECK();
            dim3 sgrid((unsigned)p.nscan);
            k_scan_blocksum<<<sgrid, block, 0, stream>>>(rowfirst, N, nd, 0, bsum);
            WFS_LAUNCH_CHECK();
            k_scan_top<<<dim3(1), block, 0, stream>>>(bsum, p.nscan, info, (long long *)m_dev, mcap);
            WFS_LAUNCH_CHECK();
            k_scan_apply<<<sgrid, block, 0, stream>>>(rowfirst, N, nd, 0, bsum, rowbase);
            WFS_LAUNCH_CHECK();
        }
    }
    if (!host_info) return WFS_OK;          // caller vouches for the indices / works with device counts: asynchronous
    long long h[4];
    WFS_HIP_CHECK(hipMemcpyAsync(h, info, sizeof(h), hipMemcpyDeviceToHost, stream));
    WFS_HIP_CHECK(hipStreamSynchronize(stream));
    WFS_REQUIRE(h[2] == 0, WFS_EINVAL, "an index row lies outside batch_size/spatial_shape");
    host_info[0] = g->subm ? N : h[0];
    host_info[1] = h[1];
    return WFS_OK;
}

extern "C" int wfs_rulebook_emit(const wfs_geometry *g, const int32_t *indices, int64_t N, int64_t M,
                                 int32_t *nbr_out, int32_t *out_indices, int32_t *nbr_in, int32_t *indice_pairs,
                                 int32_t *indice_pair_num, void *workspace, size_t workspace_bytes,
                                 const int64_t *n_dev, int32_t *overflow_dev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    (void)indices;
    WFS_REQUIRE(g, WFS_EINVAL, "NULL geometry");
    if (N == 0) {
        if (indice_pair_num) WFS_HIP_CHECK(hipMemsetAsync(indice_pair_num, 0, (size_t)g->K * 4, stream));
        return WFS_OK;
    }
    WFS_REQUIRE(nbr_out && workspace, WFS_EINVAL, "NULL device pointer");
    Plan p;
    make_plan(g, N, &p);
    WFS_REQUIRE(workspace_bytes >= p.total, WFS_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, p.total);
    WfsTimerScope timer(WFS_TIMER_RULEBOOK, stream);
    const long long *nd = (const long long *)n_dev;
    char *ws = (char *)workspace;
    long long *info = (long long *)(ws + p.off_info);
    p.tbl.keys = (int *)(ws + p.off_keys);
    dim3 grid((unsigned)wfs_cdiv(N, TB)), block(TB);
    const bool wide = !g->subm && g->K <= 32 && (long long)N * g->K < (1ll << 32);
    if (nbr_in && M > 0 && !wide) WFS_HIP_CHECK(hipMemsetAsync(nbr_in, 0xFF, (size_t)g->K * M * 4, stream));
    if (!g->subm) {
        WFS_REQUIRE(out_indices || M == 0, WFS_EINVAL, "out_indices is NULL");
        unsigned long long *ticket = (unsigned long long *)(ws + p.off_ticket);
        int *slot_id = (int *)(ws + p.off_slot_id);
        int *rowbase = (int *)(ws + p.off_rowbase);
        if (wide) {
            int *rowmask = (int *)(ws + p.off_rowfirst);
            dim3 grid2((unsigned)wfs_cdiv(N, 64), (unsigned)wfs_cdiv(g->K, 4));
            k_conv_assign2<<<grid2, block, 0, stream>>>(p.geo, N, nd, M, nbr_out, (const unsigned *)rowmask, rowbase, p.tbl,
                                                        slot_id, out_indices, info, M > 0 ? nbr_in : nullptr,
                                                        (long long)g->K * M);
            WFS_LAUNCH_CHECK();
            k_conv_finalize2<<<grid2, block, 0, stream>>>(g->K, N, nd, M, nbr_out, slot_id, nbr_in, info, overflow_dev);
            WFS_LAUNCH_CHECK();
        } else {
            k_conv_assign<<<grid, block, 0, stream>>>(p.geo, N, nd, M, nbr_out, ticket, rowbase, p.tbl, slot_id,
                                                      out_indices, info);
            WFS_LAUNCH_CHECK();
            k_conv_finalize<<<grid, block, 0, stream>>>(g->K, N, nd, M, nbr_out, slot_id, nbr_in);
            WFS_LAUNCH_CHECK();
            if (overflow_dev) {  // info[3] -> the caller's flag (sticky, as above): set if M exceeded the capacity
                k_set_if<<<1, 1, 0, stream>>>(info + 3, overflow_dev);
                WFS_LAUNCH_CHECK();
            }
        }
    } else if (nbr_in) {
        k_invert_table<<<grid, block, 0, stream>>>(g->K, N, nd, M, nbr_out, nbr_in);
        WFS_LAUNCH_CHECK();
    }
    if (g->transposed && g->K > 1) {
        // back to spconv's offset order (see offset_key); rows beyond the valid counts hold nothing anyone reads
        k_flip_rows<<<dim3((unsigned)wfs_cdiv(N, TB), (unsigned)(g->K / 2)), block, 0, stream>>>(g->K, N, nd, nbr_out);
        WFS_LAUNCH_CHECK();
        if (nbr_in && M > 0) {
            k_flip_rows<<<dim3((unsigned)wfs_cdiv(M, TB), (unsigned)(g->K / 2)), block, 0, stream>>>(g->K, M, nullptr, nbr_in);
            WFS_LAUNCH_CHECK();
        }
    }
    if (indice_pairs || indice_pair_num) {
        WFS_REQUIRE(indice_pair_num, WFS_EINVAL, "indice_pair_num is required with indice_pairs");
        int *tcount = (int *)(ws + p.off_tcount);
        k_compact_count<<<grid, block, 0, stream>>>(g->K, N, nd, p.ntiles, nbr_out, tcount);
        WFS_LAUNCH_CHECK();
        k_compact_scan<<<dim3((unsigned)g->K), block, 0, stream>>>(p.ntiles, tcount, indice_pair_num);
        WFS_LAUNCH_CHECK();
        if (indice_pairs) {
            k_compact_write<<<grid, block, 0, stream>>>(g->K, N, nd, p.ntiles, nbr_out, tcount, indice_pair_num,
                                                        indice_pairs);
            WFS_LAUNCH_CHECK();
        }
    }
    return WFS_OK;
}

// The cell -> output row map a regular conv's build leaves in its workspace (direct-grid mode, 32-bit tickets):
// ticket[cell] != 0xFFFFFFFF <=> the output cell is active, slot_id[cell] = its row.  Valid after wfs_rulebook_emit
// for as long as the workspace is neither freed nor reused.
extern "C" int wfs_rulebook_cell_map(const wfs_geometry *g, int64_t N, void *workspace, const uint32_t **ticket,
                                     const int32_t **slot_id, int64_t *cells) {
    if (!g || g->subm || N <= 0 || !workspace || !ticket || !slot_id) return 0;
    Plan p;
    make_plan(g, N, &p);
    const bool wide = g->K <= 32 && (long long)N * g->K < (1ll << 32);
    if (!p.tbl.direct || !wide) return 0;
    char *ws = (char *)workspace;
    *ticket = (const uint32_t *)(ws + p.off_ticket);
    *slot_id = (const int32_t *)(ws + p.off_slot_id);
    if (cells) *cells = p.cap;
    return 1;
}
