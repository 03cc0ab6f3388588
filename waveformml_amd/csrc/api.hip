// api.hip -- error string, geometry front door, opt-in event timing.
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "wfs_common.h"

static thread_local char g_err[512] = "";

void wfs_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int wfs_abi_version(void) { return WFS_ABI_VERSION; }
extern "C" const char *wfs_last_error(void) { return g_err; }

// Replaces the Python front door of spconv.ops.get_indice_pairs.
extern "C" int wfs_geometry_init(wfs_geometry *g) {
    WFS_REQUIRE(g != nullptr, WFS_EINVAL, "geometry is NULL");
    WFS_REQUIRE(g->ndim >= 1 && g->ndim <= WFS_MAX_DIM, WFS_EINVAL, "ndim %d not in [1,%d]", g->ndim,
                WFS_MAX_DIM);
    WFS_REQUIRE(g->batch_size >= 0, WFS_EINVAL, "negative batch_size");
    WFS_REQUIRE(!(g->subm && g->transposed), WFS_EINVAL, "a submanifold convolution cannot be transposed");
    int64_t K = 1, vol = 1;
    for (int i = 0; i < g->ndim; ++i) {
        WFS_REQUIRE(g->ksize[i] >= 1 && g->stride[i] >= 1 && g->dilation[i] >= 1 && g->padding[i] >= 0 &&
                        g->spatial[i] >= 1,
                    WFS_EINVAL, "bad conv parameter in dim %d", i);
        WFS_REQUIRE(g->stride[i] == 1 || g->dilation[i] == 1, WFS_EINVAL,
                    "stride>1 together with dilation>1 is not supported (dim %d)", i);
        if (g->subm) {
            g->stride[i] = 1;
            g->padding[i] = g->ksize[i] / 2;
            g->out_shape[i] = g->spatial[i];
        } else if (g->transposed) {
            WFS_REQUIRE(g->output_padding[i] >= 0, WFS_EINVAL, "negative output_padding in dim %d", i);
            const int64_t o = ((int64_t)g->spatial[i] - 1) * g->stride[i] - 2 * (int64_t)g->padding[i] + g->ksize[i] +
                              g->output_padding[i];            // spconv's get_deconv_output_size (no dilation term)
            WFS_REQUIRE(o >= 1 && o < ((int64_t)1 << 31), WFS_EINVAL, "deconv output size %lld in dim %d", (long long)o, i);
            g->out_shape[i] = (int32_t)o;
        } else {
            int64_t o = (int64_t)g->spatial[i] + 2 * g->padding[i] - (int64_t)g->dilation[i] * (g->ksize[i] - 1) - 1;
            // python floor division, as spconv's get_conv_output_size
            o = (o >= 0 ? o / g->stride[i] : -((-o + g->stride[i] - 1) / g->stride[i])) + 1;
            WFS_REQUIRE(o >= 1, WFS_EINVAL, "conv output size %lld < 1 in dim %d", (long long)o, i);
            g->out_shape[i] = (int32_t)o;
        }
        K *= g->ksize[i];
        vol *= g->out_shape[i];
        WFS_REQUIRE(K <= (1 << 20), WFS_EINVAL, "kernel volume too large");
    }
    for (int i = g->ndim; i < WFS_MAX_DIM; ++i) {
        g->spatial[i] = g->out_shape[i] = g->ksize[i] = g->stride[i] = g->dilation[i] = 1;
        g->padding[i] = 0;
    }
    g->K = (int32_t)K;
    if ((int64_t)g->batch_size * vol >= ((int64_t)1 << 31)) {
        wfs_set_error("batch_size * prod(out_shape) = %lld >= 2^31", (long long)((int64_t)g->batch_size * vol));
        return WFS_EOVERFLOW;
    }
    return WFS_OK;
}

// ---------------------------------------------------------------- event timing
struct TimerRec {
    int timer;
    hipEvent_t a, b;
};
static std::mutex g_tmu;
static bool g_timing = false;
static std::vector<TimerRec *> g_recs;

WfsTimerScope::WfsTimerScope(int timer_, hipStream_t stream_) : timer(timer_), stream(stream_), rec(nullptr) {
    if (!g_timing) return;
    TimerRec *r = new TimerRec;
    r->timer = timer;
    if (hipEventCreate(&r->a) != hipSuccess || hipEventCreate(&r->b) != hipSuccess) {
        delete r;
        return;
    }
    (void)hipEventRecord(r->a, stream);
    rec = r;
}
WfsTimerScope::~WfsTimerScope() {
    if (!rec) return;
    TimerRec *r = (TimerRec *)rec;
    (void)hipEventRecord(r->b, stream);
    std::lock_guard<std::mutex> lk(g_tmu);
    g_recs.push_back(r);
}

extern "C" int wfs_timing_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(g_tmu);
    for (TimerRec *r : g_recs) {
        (void)hipEventDestroy(r->a);
        (void)hipEventDestroy(r->b);
        delete r;
    }
    g_recs.clear();
    g_timing = on != 0;
    return WFS_OK;
}

extern "C" int wfs_timing_read(int32_t timer, double *total_ms, int64_t *launches) {
    std::lock_guard<std::mutex> lk(g_tmu);
    double tot = 0;
    int64_t n = 0;
    for (TimerRec *r : g_recs) {
        if (r->timer != timer) continue;
        WFS_HIP_CHECK(hipEventSynchronize(r->b));
        float ms = 0;
        WFS_HIP_CHECK(hipEventElapsedTime(&ms, r->a, r->b));
        tot += ms;
        ++n;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return WFS_OK;
}
