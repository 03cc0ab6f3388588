// optim.hip -- the optimizer update of the PSD training step on ONE flat parameter buffer, in one launch.
//
// The reference builds its optimizer from the config (src/engineering/LitPSD.py:60-76; the examples use
// torch.optim.SGD with momentum 0.98 and nesterov, config/examples/GEP.json:51-69).  torch's single-tensor SGD is four
// elementwise launches (and its "fused" multi-tensor kernel runs a single flat tensor on a handful of blocks); at the
// PSD batch sizes every launch costs ~5 us, so the update is one kernel here.  Same arithmetic, in torch's order
// (torch/optim/sgd.py _single_tensor_sgd):
//     g = grad + weight_decay * p
//     buf = g                                   (first step)      buf = momentum * buf + (1 - dampening) * g
//     g = nesterov ? g + momentum * buf : buf
//     p -= lr * g
#include "wfs_common.h"

namespace {

__global__ void __launch_bounds__(256) k_sgd_step(float *__restrict__ P, const float *__restrict__ G,
                                                  float *__restrict__ M, long long n, const float *__restrict__ lr_dev,
                                                  float momentum, float dampening, float weight_decay, int nesterov,
                                                  int first_step) {
    const float lr = *lr_dev;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float p = P[i];
        float g = G[i];
        if (weight_decay != 0.f) g = fmaf(weight_decay, p, g);
        if (M) {
            float b = first_step ? g : fmaf(1.f - dampening, g, M[i] * momentum);
            M[i] = b;
            g = nesterov ? fmaf(momentum, b, g) : b;
        }
        P[i] = fmaf(-lr, g, p);
    }
}

}  // namespace

extern "C" int wfs_sgd_step(float *param, const float *grad, float *momentum_buf, int64_t n, const float *lr_dev,
                            float momentum, float dampening, float weight_decay, int32_t nesterov, int32_t first_step,
                            void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n == 0) return WFS_OK;
    WFS_REQUIRE(param && grad && lr_dev, WFS_EINVAL, "NULL device pointer");
    WFS_REQUIRE(momentum_buf || momentum == 0.f, WFS_EINVAL, "momentum needs a buffer");
    long long blocks = wfs_cdiv(n, 256);
    if (blocks > 2048) blocks = 2048;
    k_sgd_step<<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(param, grad, momentum != 0.f ? momentum_buf : nullptr, n,
                                                                 lr_dev, momentum, dampening, weight_decay, nesterov,
                                                                 first_step);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Hand-over of a batch to a captured step in ONE launch: a replayed HIP graph reads its inputs from fixed buffers, so
// every step starts by copying the batch there -- coordinates, features, labels, the row count -- and the reference's
// forward then permutes the coordinate columns to batch-first (src/models/SPConvNet.py:64).  Five launches (three
// copies, a fill, an index kernel) at ~5 us each become one.
namespace {

struct Perm {
    int v[8];
};

__global__ void __launch_bounds__(256) k_load_batch(const int *__restrict__ coords, long long n, int cols, Perm perm,
                                                    int *__restrict__ coords_dst, int *__restrict__ indices_dst,
                                                    const uint4 *__restrict__ feats, uint4 *__restrict__ feats_dst,
                                                    long long feat_words, const unsigned char *__restrict__ feats_tail,
                                                    unsigned char *__restrict__ feats_tail_dst, int tail_bytes,
                                                    const long long *__restrict__ labels, long long *__restrict__ labels_dst,
                                                    long long B, long long *__restrict__ n_valid_dst,
                                                    int *__restrict__ ev_off, int ev_B, int ev_col) {
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, nth = (long long)gridDim.x * 256;
    if (tid == 0 && n_valid_dst) *n_valid_dst = n;
    // event offsets of the batch (evrulebook.hip k_event_offsets, same words, same flags): the first WFS_EVENT_FLAG_WORDS
    // blocks walk the batch column of the SOURCE rows -- the captured step then starts with the rulebook build itself
    if (ev_off && blockIdx.x < WFS_EVENT_FLAG_WORDS) {
        int bad = 0;
        if (n == 0)
            for (int e = blockIdx.x * 256 + threadIdx.x; e <= ev_B; e += WFS_EVENT_FLAG_WORDS * 256) ev_off[e] = 0;
        for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < n; j += (long long)WFS_EVENT_FLAG_WORDS * 256) {
            const int b = coords[j * cols + ev_col];
            const int bp = j > 0 ? coords[(j - 1) * cols + ev_col] : -1;
            const bool ok = b >= 0 && b < ev_B && b >= bp && bp >= -1 && bp < ev_B;
            bad |= ok ? 0 : 1;
            if (ok) {
                for (int e = bp + 1; e <= b; ++e) ev_off[e] = (int)j;
                if (j == n - 1)
                    for (int e = b + 1; e <= ev_B; ++e) ev_off[e] = (int)n;
            }
        }
        bad = __syncthreads_or(bad);
        if (threadIdx.x == 0) ev_off[ev_B + 1 + blockIdx.x] = bad;
    }
    if (cols == 4 && ((reinterpret_cast<uintptr_t>(coords) | reinterpret_cast<uintptr_t>(coords_dst) |
                       reinterpret_cast<uintptr_t>(indices_dst)) & 15) == 0) {
        // the PSD nets' rows (x, y, t, event): one 16-byte row per thread, no division
        const int4 *src = reinterpret_cast<const int4 *>(coords);
        int4 *dst = reinterpret_cast<int4 *>(coords_dst), *idx = reinterpret_cast<int4 *>(indices_dst);
        const int p0 = perm.v[0], p1 = perm.v[1], p2 = perm.v[2], p3 = perm.v[3];
        auto pick = [](const int4 &r, int p) { return p == 0 ? r.x : (p == 1 ? r.y : (p == 2 ? r.z : r.w)); };
        for (long long row = tid; row < n; row += nth) {
            const int4 r = src[row];
            if (dst) dst[row] = r;
            if (idx) idx[row] = int4{pick(r, p0), pick(r, p1), pick(r, p2), pick(r, p3)};
        }
    } else {
        for (long long i = tid; i < n * cols; i += nth) {
            const long long row = i / cols;
            const int col = (int)(i - row * cols);
            if (coords_dst) coords_dst[i] = coords[i];
            if (indices_dst) indices_dst[i] = coords[row * cols + perm.v[col]];
        }
    }
    for (long long i = tid; i < feat_words; i += nth) feats_dst[i] = feats[i];
    for (long long i = tid; i < tail_bytes; i += nth) feats_tail_dst[i] = feats_tail[i];
    for (long long i = tid; i < B; i += nth) labels_dst[i] = labels[i];
}

}  // namespace

extern "C" int wfs_load_batch(const int32_t *coords, int64_t n, int32_t cols, const int32_t *perm_host,
                              int32_t *coords_dst, int32_t *indices_dst, const void *feats, void *feats_dst,
                              int64_t feat_bytes, const int64_t *labels, int64_t *labels_dst, int64_t B,
                              int64_t *n_valid_dst, int32_t *event_offsets, int32_t events, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WFS_REQUIRE(cols >= 1 && cols <= 8, WFS_EINVAL, "bad coordinate width %d", cols);
    WFS_REQUIRE(n >= 0 && B >= 0 && feat_bytes >= 0, WFS_EINVAL, "negative size");
    WFS_REQUIRE((n == 0 || coords) && (feat_bytes == 0 || (feats && feats_dst)) && (B == 0 || (labels && labels_dst)),
                WFS_EINVAL, "NULL device pointer");
    Perm pm;
    for (int c = 0; c < 8; ++c) {
        pm.v[c] = (perm_host && c < cols) ? perm_host[c] : c;
        WFS_REQUIRE(pm.v[c] >= 0 && pm.v[c] < (c < cols ? cols : 8), WFS_EINVAL, "bad column permutation");
    }
    const bool aligned = ((uintptr_t)feats % 16 == 0) && ((uintptr_t)feats_dst % 16 == 0);
    const long long words = aligned ? feat_bytes / 16 : 0;
    const int tail = (int)(feat_bytes - words * 16 > (1 << 30) ? 0 : feat_bytes - words * 16);
    WFS_REQUIRE(aligned || feat_bytes < (1 << 30), WFS_EINVAL, "unaligned feature buffers");
    long long work = n * cols > words ? n * cols : words;
    long long blocks = wfs_cdiv(work > 0 ? work : 1, 256);
    if (blocks > 1024) blocks = 1024;
    WFS_REQUIRE(!event_offsets || events >= 1, WFS_EINVAL, "event offsets of %d events", events);
    if (event_offsets && blocks < WFS_EVENT_FLAG_WORDS) blocks = WFS_EVENT_FLAG_WORDS;
    k_load_batch<<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(
        coords, n, cols, pm, coords_dst, indices_dst, (const uint4 *)feats, (uint4 *)feats_dst, words,
        (const unsigned char *)feats + words * 16, (unsigned char *)feats_dst + words * 16, tail, (const long long *)labels,
        (long long *)labels_dst, B, (long long *)n_valid_dst, event_offsets, events, pm.v[0]);
    WFS_LAUNCH_CHECK();
    return WFS_OK;
}
