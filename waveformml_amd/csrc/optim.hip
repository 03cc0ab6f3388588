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
