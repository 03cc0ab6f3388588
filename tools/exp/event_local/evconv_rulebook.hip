// evconv_rulebook.hip -- event-local REGULAR / STRIDED conv rulebook build (round 3 experiment; NOT part of libwfsparse).
// Bit-identical tables to wfs_rulebook_plan + _emit (first-seen numbering), measured slower: 59 vs 52 us at
// the PSD batch -- one workgroup per event is bound by the largest event (3x the mean).  See README.md here.
#include <stdlib.h>

#include "wfs_common.h"
#include "event_local.h"

namespace {

struct EGeo {
    int ndim, K;
    int spatial[4], out_shape[4], ksize[4], stride[4], padding[4], dilation[4];
};
#ifndef ER_KNOCK
#define ER_KNOCK 0
#endif
constexpr int ER_THREADS = 512;

__device__ __forceinline__ long long valid_rows(long long R, const long long *r_dev) {
    long long v = r_dev ? *r_dev : R;
    return v < R ? v : R;
}

__device__ __forceinline__ bool ev_structured(const int *ev, int B) {
    const int fl = ev[B + 1 + (threadIdx.x & 63)];
    return __ballot(fl != 0) == 0ull;
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

}  // namespace

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
