/*
 * oracle/spconv_ref.c -- TEST INFRASTRUCTURE ONLY.  Not shipped, not linked by the product.
 *
 * Plain-C, single-threaded CPU restatement of the sparse-convolution arithmetic that
 * BlaineHeffron/WaveformML executes through the third-party package spconv~=1.2.1
 * (reference requirements.txt:15; call sites src/models/SPConvBlocks.py:75,134,498,803-810,
 * src/models/SPConvNet.py:23,64, src/engineering/LitBase.py:138-146).
 *
 * spconv is NOT in /root/reference (not vendored, not a submodule) and cannot be installed
 * offline, and the reference has no tests (reference .gitignore:136-137), so there are no
 * golden vectors for this path:   **PARITY UNPINNED**   against the upstream binary.
 * What this file follows is the published v1.2.1 algorithm as restated in SURVEY.md
 * Appendix A (A.2 front door, A.3 CPU rulebook, A.4 indice_conv / backward, A.1 dense()).
 * Values are pinned independently by tests/test_oracle.py against dense
 * torch.nn.functional.conv{2,3}d on the densified input (the densify-then-conv pattern the
 * reference itself uses in src/models/DenseConvNet.py:26-34); rulebook ORDER is pinned only
 * by the appendix and by the hand-enumerated fixtures in tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define WFO_MAX_DIM 4

/* ------------------------------------------------------------------ tiny int64->int32 map */
typedef struct {
    int64_t *keys;
    int32_t *vals;
    uint64_t cap; /* power of two */
} wfo_map;

static int wfo_map_init(wfo_map *m, int64_t n_expected) {
    uint64_t cap = 16;
    while (cap < (uint64_t)(2 * n_expected + 2)) cap <<= 1;
    m->cap = cap;
    m->keys = (int64_t *)malloc(cap * sizeof(int64_t));
    m->vals = (int32_t *)malloc(cap * sizeof(int32_t));
    if (!m->keys || !m->vals) return -1;
    for (uint64_t i = 0; i < cap; ++i) m->keys[i] = -1;
    return 0;
}
static void wfo_map_free(wfo_map *m) {
    free(m->keys);
    free(m->vals);
}
static uint64_t wfo_mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
/* returns slot of key, or slot where it would be inserted */
static uint64_t wfo_map_slot(const wfo_map *m, int64_t key) {
    uint64_t s = wfo_mix((uint64_t)key) & (m->cap - 1);
    while (m->keys[s] != -1 && m->keys[s] != key) s = (s + 1) & (m->cap - 1);
    return s;
}

/* ------------------------------------------------------------------ geometry (Appendix A.3)
 * getValidOutPos: per dim lo=(x-(k-1)d-1+s+p)/s, hi=(x+p)/s with C truncating division,
 * cnt=(hi-lo)/d+1; Cartesian product enumerated with the LAST dim fastest, candidate
 * out_j = hi_j - counter_j*d_j; kernel offset = sum_j m_j*((x_j-out_j*s_j+p_j)/d_j) with m the
 * row-major strides of the kernel; kept iff 0<=out_j<out_shape_j for every j.
 * out[] receives kept candidates as (out_0..out_{D-1}, offset) records.                      */
static int wfo_valid_out_pos(const int32_t *in_pos, int ndim, const int32_t *ksize,
                             const int32_t *stride, const int32_t *padding,
                             const int32_t *dilation, const int32_t *out_shape, int32_t *out) {
    int32_t lowers[WFO_MAX_DIM], uppers[WFO_MAX_DIM], counter[WFO_MAX_DIM], csize[WFO_MAX_DIM];
    int32_t num_points = 1;
    int point_counter = 0;
    for (int i = 0; i < ndim; ++i) {
        lowers[i] = (in_pos[i] - (ksize[i] - 1) * dilation[i] - 1 + stride[i] + padding[i]) / stride[i];
        uppers[i] = (in_pos[i] + padding[i]) / stride[i];
    }
    for (int i = 0; i < ndim; ++i) {
        csize[i] = (uppers[i] - lowers[i]) / dilation[i] + 1;
        num_points *= csize[i];
        counter[i] = 0;
    }
    for (int32_t i = 0; i < num_points; ++i) {
        int valid = 1;
        int32_t m = 1, offset = 0;
        for (int j = ndim - 1; j >= 0; --j) {
            int32_t val = uppers[j] - counter[j] * dilation[j];
            out[point_counter * (ndim + 1) + j] = val;
            if (val < 0 || val > out_shape[j] - 1) valid = 0;
            offset += m * (in_pos[j] - val * stride[j] + padding[j]) / dilation[j];
            m *= ksize[j];
        }
        out[point_counter * (ndim + 1) + ndim] = offset;
        if (valid) ++point_counter;
        counter[ndim - 1] += 1;
        for (int c = ndim - 1; c >= 0; --c) {
            if (counter[c] == csize[c] && c > 0) {
                counter[c - 1] += 1;
                counter[c] = 0;
            }
        }
    }
    return point_counter;
}

/* getValidOutPosTranspose (spconv 1.2.1 include/spconv/geometry.h, restated from memory of upstream: the source is not
 * in this container -- PARITY UNPINNED on the enumeration order): per dim lower = x*s - p, upper = lower + (k-1)*d;
 * candidates val_j = upper_j - counter_j*d_j, LAST dim fastest, i.e. from the highest kernel offset down;
 * offset = sum_j m_j*((val_j - lower_j)/d_j) with m the row-major strides of the kernel; kept iff inside out_shape.  */
static int wfo_valid_out_pos_transpose(const int32_t *in_pos, int ndim, const int32_t *ksize,
                                       const int32_t *stride, const int32_t *padding,
                                       const int32_t *dilation, const int32_t *out_shape, int32_t *out) {
    int32_t lowers[WFO_MAX_DIM], uppers[WFO_MAX_DIM], counter[WFO_MAX_DIM], csize[WFO_MAX_DIM];
    int32_t num_points = 1;
    int point_counter = 0;
    for (int i = 0; i < ndim; ++i) {
        lowers[i] = in_pos[i] * stride[i] - padding[i];
        uppers[i] = lowers[i] + (ksize[i] - 1) * dilation[i];
    }
    for (int i = 0; i < ndim; ++i) {
        csize[i] = (uppers[i] - lowers[i]) / dilation[i] + 1;
        num_points *= csize[i];
        counter[i] = 0;
    }
    for (int32_t i = 0; i < num_points; ++i) {
        int valid = 1;
        int32_t m = 1, offset = 0;
        for (int j = ndim - 1; j >= 0; --j) {
            int32_t val = uppers[j] - counter[j] * dilation[j];
            out[point_counter * (ndim + 1) + j] = val;
            if (val < 0 || val > out_shape[j] - 1) valid = 0;
            offset += m * (val - lowers[j]) / dilation[j];
            m *= ksize[j];
        }
        out[point_counter * (ndim + 1) + ndim] = offset;
        if (valid) ++point_counter;
        counter[ndim - 1] += 1;
        for (int c = ndim - 1; c >= 0; --c) {
            if (counter[c] == csize[c] && c > 0) {
                counter[c - 1] += 1;
                counter[c] = 0;
            }
        }
    }
    return point_counter;
}

static int64_t wfo_row_major(const int32_t *pos, int ndim, const int32_t *shape) {
    int64_t idx = 0;
    for (int i = 0; i < ndim; ++i) idx = idx * shape[i] + pos[i];
    return idx;
}

/* ------------------------------------------------------------------ rulebooks
 * indices     int32 [N, ndim+1]  (batch, x0..x_{D-1})
 * pairs       int32 [2, K, N]    pre-filled with -1 by the caller (A.2)
 * pair_num    int32 [K]          pre-zeroed by the caller
 * SubM (A.3): pass 1 hash[key(in_j)] = j (duplicates: last wins); pass 2 for every j, every kept
 * candidate in enumeration order, if the hash holds the key append (j, hash[key]).
 * The SubM front door (A.2) forces padding = ksize/2 and stride = 1, out_shape = spatial.      */
int wfo_rulebook_subm(const int32_t *indices, int64_t N, int ndim, const int32_t *spatial,
                      const int32_t *ksize, const int32_t *dilation, int32_t *pairs,
                      int32_t *pair_num) {
    if (ndim < 1 || ndim > WFO_MAX_DIM) return -1;
    int32_t stride[WFO_MAX_DIM], padding[WFO_MAX_DIM];
    int64_t K = 1, volume = 1;
    for (int i = 0; i < ndim; ++i) {
        stride[i] = 1;
        padding[i] = ksize[i] / 2;
        K *= ksize[i];
        volume *= spatial[i];
    }
    wfo_map map;
    if (wfo_map_init(&map, N)) return -2;
    for (int64_t j = 0; j < N; ++j) {
        const int32_t *row = indices + j * (ndim + 1);
        int64_t key = wfo_row_major(row + 1, ndim, spatial) + volume * (int64_t)row[0];
        uint64_t s = wfo_map_slot(&map, key);
        map.keys[s] = key;
        map.vals[s] = (int32_t)j;
    }
    int32_t *pts = (int32_t *)malloc((size_t)K * (ndim + 1) * sizeof(int32_t));
    for (int64_t j = 0; j < N; ++j) {
        const int32_t *row = indices + j * (ndim + 1);
        int n = wfo_valid_out_pos(row + 1, ndim, ksize, stride, padding, dilation, spatial, pts);
        for (int i = 0; i < n; ++i) {
            const int32_t *p = pts + i * (ndim + 1);
            int32_t off = p[ndim];
            int64_t key = wfo_row_major(p, ndim, spatial) + volume * (int64_t)row[0];
            uint64_t s = wfo_map_slot(&map, key);
            if (map.keys[s] == key) {
                int32_t c = pair_num[off]++;
                pairs[(0 * K + off) * N + c] = (int32_t)j;
                pairs[(1 * K + off) * N + c] = map.vals[s];
            }
        }
    }
    free(pts);
    wfo_map_free(&map);
    return 0;
}

/* Regular / strided conv (A.3): sequentially for input row j, for each kept candidate in
 * enumeration order: unseen key -> next output id (first-seen numbering), record
 * out_indices[id] = (batch, out_pos); append (j, id) to that offset.
 * out_indices int32 [>= M, ndim+1] (capacity N*K rows is always enough). Returns M (or <0). */
static int64_t wfo_rulebook_conv_impl(const int32_t *indices, int64_t N, int ndim, const int32_t *out_shape,
                                      const int32_t *ksize, const int32_t *stride, const int32_t *padding,
                                      const int32_t *dilation, int32_t *out_indices, int32_t *pairs,
                                      int32_t *pair_num, int transpose);

int64_t wfo_rulebook_conv(const int32_t *indices, int64_t N, int ndim, const int32_t *out_shape,
                          const int32_t *ksize, const int32_t *stride, const int32_t *padding,
                          const int32_t *dilation, int32_t *out_indices, int32_t *pairs,
                          int32_t *pair_num) {
    return wfo_rulebook_conv_impl(indices, N, ndim, out_shape, ksize, stride, padding, dilation, out_indices, pairs,
                                  pair_num, 0);
}

/* Transposed conv (SparseConvTranspose): the same sequential first-seen numbering over getValidOutPosTranspose. */
int64_t wfo_rulebook_conv_transpose(const int32_t *indices, int64_t N, int ndim, const int32_t *out_shape,
                                    const int32_t *ksize, const int32_t *stride, const int32_t *padding,
                                    const int32_t *dilation, int32_t *out_indices, int32_t *pairs,
                                    int32_t *pair_num) {
    return wfo_rulebook_conv_impl(indices, N, ndim, out_shape, ksize, stride, padding, dilation, out_indices, pairs,
                                  pair_num, 1);
}

static int64_t wfo_rulebook_conv_impl(const int32_t *indices, int64_t N, int ndim, const int32_t *out_shape,
                                      const int32_t *ksize, const int32_t *stride, const int32_t *padding,
                                      const int32_t *dilation, int32_t *out_indices, int32_t *pairs,
                                      int32_t *pair_num, int transpose) {
    if (ndim < 1 || ndim > WFO_MAX_DIM) return -1;
    int64_t K = 1, volume = 1;
    for (int i = 0; i < ndim; ++i) {
        K *= ksize[i];
        volume *= out_shape[i];
    }
    wfo_map map;
    if (wfo_map_init(&map, N * K)) return -2;
    int32_t *pts = (int32_t *)malloc((size_t)K * (ndim + 1) * sizeof(int32_t));
    int64_t M = 0;
    for (int64_t j = 0; j < N; ++j) {
        const int32_t *row = indices + j * (ndim + 1);
        int n = transpose ? wfo_valid_out_pos_transpose(row + 1, ndim, ksize, stride, padding, dilation, out_shape, pts)
                          : wfo_valid_out_pos(row + 1, ndim, ksize, stride, padding, dilation, out_shape, pts);
        for (int i = 0; i < n; ++i) {
            const int32_t *p = pts + i * (ndim + 1);
            int32_t off = p[ndim];
            int64_t key = wfo_row_major(p, ndim, out_shape) + volume * (int64_t)row[0];
            uint64_t s = wfo_map_slot(&map, key);
            int32_t id;
            if (map.keys[s] != key) {
                id = (int32_t)M++;
                map.keys[s] = key;
                map.vals[s] = id;
                out_indices[(int64_t)id * (ndim + 1)] = row[0];
                for (int d = 0; d < ndim; ++d) out_indices[(int64_t)id * (ndim + 1) + 1 + d] = p[d];
            } else {
                id = map.vals[s];
            }
            int32_t c = pair_num[off]++;
            pairs[(0 * K + off) * N + c] = (int32_t)j;
            pairs[(1 * K + off) * N + c] = id;
        }
    }
    free(pts);
    wfo_map_free(&map);
    return M;
}

/* ------------------------------------------------------------------ indice_conv (A.4, Native)
 * features [n_in, Cin], filters [K, Cin, Cout], out [n_out, Cout] (overwritten).
 * subm: out = X . W[k*], k* = first argmax of pair_num, that offset skipped in the loop.
 * inverse: the roles of the two pair rows are swapped.
 * Per offset: gather rows -> buf = rows . W[k] -> out[dst] += buf, in pair order.             */
static int wfo_argmax(const int32_t *v, int64_t K) {
    int best = 0;
    for (int64_t k = 1; k < K; ++k)
        if (v[k] > v[best]) best = (int)k;
    return best;
}

int wfo_indice_conv_fwd(const float *features, const float *filters, const int32_t *pairs,
                        const int32_t *pair_num, int64_t n_in, int64_t n_out, int64_t K,
                        int64_t pair_cap, int Cin, int Cout, int inverse, int subm, float *out) {
    memset(out, 0, (size_t)n_out * Cout * sizeof(float));
    int kmax = wfo_argmax(pair_num, K);
    float *buf = (float *)malloc((size_t)Cout * sizeof(float));
    if (subm) {
        const float *W = filters + (int64_t)kmax * Cin * Cout;
        for (int64_t r = 0; r < n_in && r < n_out; ++r) {
            float *o = out + r * Cout;
            const float *x = features + r * Cin;
            for (int ci = 0; ci < Cin; ++ci) {
                float xv = x[ci];
                const float *w = W + (int64_t)ci * Cout;
                for (int co = 0; co < Cout; ++co) o[co] += xv * w[co];
            }
        }
    }
    const int32_t *src_rows = pairs + (inverse ? 1 : 0) * K * pair_cap;
    const int32_t *dst_rows = pairs + (inverse ? 0 : 1) * K * pair_cap;
    for (int64_t k = 0; k < K; ++k) {
        int32_t n = pair_num[k];
        if (n <= 0 || (subm && k == kmax)) continue;
        const float *W = filters + k * Cin * Cout;
        for (int32_t p = 0; p < n; ++p) {
            const float *x = features + (int64_t)src_rows[k * pair_cap + p] * Cin;
            float *o = out + (int64_t)dst_rows[k * pair_cap + p] * Cout;
            for (int co = 0; co < Cout; ++co) buf[co] = 0.f;
            for (int ci = 0; ci < Cin; ++ci) {
                float xv = x[ci];
                const float *w = W + (int64_t)ci * Cout;
                for (int co = 0; co < Cout; ++co) buf[co] += xv * w[co];
            }
            for (int co = 0; co < Cout; ++co) o[co] += buf[co];
        }
    }
    free(buf);
    return 0;
}

/* backward (A.4): dW = zeros, dX = zeros; subm centre: dW[k*] = X^T dY, dX = dY W[k*]^T;
 * per k: bi = gather(X), bo = gather(dY); dW[k] = bi^T bo; dX[in] += bo W[k]^T.              */
int wfo_indice_conv_bwd(const float *features, const float *filters, const float *dout,
                        const int32_t *pairs, const int32_t *pair_num, int64_t n_in,
                        int64_t n_out, int64_t K, int64_t pair_cap, int Cin, int Cout,
                        int inverse, int subm, float *din, float *dfilters) {
    memset(din, 0, (size_t)n_in * Cin * sizeof(float));
    memset(dfilters, 0, (size_t)K * Cin * Cout * sizeof(float));
    int kmax = wfo_argmax(pair_num, K);
    if (subm) {
        const float *W = filters + (int64_t)kmax * Cin * Cout;
        float *dW = dfilters + (int64_t)kmax * Cin * Cout;
        for (int64_t r = 0; r < n_in && r < n_out; ++r) {
            const float *x = features + r * Cin;
            const float *g = dout + r * Cout;
            float *dx = din + r * Cin;
            for (int ci = 0; ci < Cin; ++ci) {
                float acc = 0.f;
                for (int co = 0; co < Cout; ++co) {
                    dW[(int64_t)ci * Cout + co] += x[ci] * g[co];
                    acc += g[co] * W[(int64_t)ci * Cout + co];
                }
                dx[ci] += acc;
            }
        }
    }
    const int32_t *src_rows = pairs + (inverse ? 1 : 0) * K * pair_cap;
    const int32_t *dst_rows = pairs + (inverse ? 0 : 1) * K * pair_cap;
    for (int64_t k = 0; k < K; ++k) {
        int32_t n = pair_num[k];
        if (n <= 0 || (subm && k == kmax)) continue;
        const float *W = filters + k * Cin * Cout;
        float *dW = dfilters + k * Cin * Cout;
        for (int32_t p = 0; p < n; ++p) {
            int64_t ri = src_rows[k * pair_cap + p], ro = dst_rows[k * pair_cap + p];
            const float *x = features + ri * Cin;
            const float *g = dout + ro * Cout;
            float *dx = din + ri * Cin;
            for (int ci = 0; ci < Cin; ++ci) {
                float acc = 0.f;
                for (int co = 0; co < Cout; ++co) {
                    dW[(int64_t)ci * Cout + co] += x[ci] * g[co];
                    acc += g[co] * W[(int64_t)ci * Cout + co];
                }
                dx[ci] += acc;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ SparseConvTensor.dense() (A.1)
 * out = zeros([B, *spatial, C]); out[idx] = features (assignment in row order, last wins);
 * returned channels-first [B, C, *spatial] contiguous.                                        */
int wfo_to_dense(const float *features, const int32_t *indices, int64_t M, int ndim,
                 const int32_t *spatial, int batch_size, int C, float *out) {
    int64_t volume = 1;
    for (int i = 0; i < ndim; ++i) volume *= spatial[i];
    memset(out, 0, (size_t)batch_size * volume * C * sizeof(float));
    for (int64_t m = 0; m < M; ++m) {
        const int32_t *row = indices + m * (ndim + 1);
        int64_t pos = wfo_row_major(row + 1, ndim, spatial);
        for (int c = 0; c < C; ++c) out[((int64_t)row[0] * C + c) * volume + pos] = features[m * C + c];
    }
    return 0;
}
