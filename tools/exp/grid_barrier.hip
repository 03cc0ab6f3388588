// grid_barrier.hip -- what a grid-wide barrier costs inside one kernel on this GPU (256 workgroups x 256 threads, one per
// compute unit), against splitting the kernel in two graph nodes.  Variants: fences on / off, how the waiters poll.
// build: hipcc --offload-arch=gfx950 -O2 tools/exp/grid_barrier.hip -o tools/exp/grid_barrier ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// mode bits: 1 = fences, 2 = sleep between polls, 4 = poll with an atomic RMW (add 0) instead of a load,
//            8 = every wave's lane 0 does NOT poll, only block thread 0 (always), 16 = sc1 load via builtin
template <int MODE>
__global__ void __launch_bounds__(256) k_bar(unsigned int *bar, float *part, float *out, int nb) {
    // phase 1: every block writes a partial
    if (threadIdx.x < 64) part[blockIdx.x * 64 + threadIdx.x] = (float)(blockIdx.x + threadIdx.x);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE & 1) __threadfence();
        atomicAdd(&bar[0], 1u);
        unsigned int tries = 0;
        for (;;) {
            unsigned int v;
            if (MODE & 4) v = atomicAdd(&bar[0], 0u);
            else v = __hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v >= (unsigned)nb || ++tries > (1u << 20)) break;
            if (MODE & 2) __builtin_amdgcn_s_sleep(2);
        }
        if (MODE & 1) __threadfence();
        if (atomicAdd(&bar[1], 1u) == (unsigned)nb - 1) {
            __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&bar[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    // phase 2: every block folds all partials
    float a = 0.f;
    for (int b = threadIdx.x / 64; b < nb; b += 4) a += part[b * 64 + (threadIdx.x & 63)];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}
__global__ void __launch_bounds__(256) k_p1(float *part) {
    if (threadIdx.x < 64) part[blockIdx.x * 64 + threadIdx.x] = (float)(blockIdx.x + threadIdx.x);
}
__global__ void __launch_bounds__(256) k_p2(const float *part, float *out, int nb) {
    float a = 0.f;
    for (int b = threadIdx.x / 64; b < nb; b += 4) a += part[b * 64 + (threadIdx.x & 63)];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    unsigned int *bar;
    float *part, *out;
    const int nb = 256;
    CK(hipMalloc((void **)&bar, 64));
    CK(hipMemset(bar, 0, 64));
    CK(hipMalloc((void **)&part, nb * 64 * 4));
    CK(hipMalloc((void **)&out, nb * 256 * 4));
    const int nodes = 50, reps = 40;
    auto time_graph = [&](auto launch, const char *what) -> int {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < nodes; ++i) launch(i);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        CK(hipEventRecord(a, s));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        unsigned int h[2];
        CK(hipMemcpy(h, bar, 8, hipMemcpyDeviceToHost));
        printf("%-64s %6.2f us per step   (barrier words after: %u %u)\n", what, ms * 1e3 / (reps * nodes), h[0], h[1]);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
        return 0;
    };
    if (time_graph([&](int) { k_p1<<<nb, 256, 0, s>>>(part); k_p2<<<nb, 256, 0, s>>>(part, out, nb); }, "two kernels (write partials | fold)")) return 1;
    if (time_graph([&](int) { k_bar<0><<<nb, 256, 0, s>>>(bar, part, out, nb); }, "barrier: no fences, load poll")) return 1;
    if (time_graph([&](int) { k_bar<1><<<nb, 256, 0, s>>>(bar, part, out, nb); }, "barrier: fences, load poll")) return 1;
    if (time_graph([&](int) { k_bar<3><<<nb, 256, 0, s>>>(bar, part, out, nb); }, "barrier: fences, load poll + sleep")) return 1;
    if (time_graph([&](int) { k_bar<5><<<nb, 256, 0, s>>>(bar, part, out, nb); }, "barrier: fences, atomic poll")) return 1;
    if (time_graph([&](int) { k_bar<7><<<nb, 256, 0, s>>>(bar, part, out, nb); }, "barrier: fences, atomic poll + sleep")) return 1;
    if (time_graph([&](int) { k_bar<4><<<nb, 256, 0, s>>>(bar, part, out, nb); }, "barrier: no fences, atomic poll")) return 1;
    for (int n2 : {64, 128}) {
        char w[96];
        snprintf(w, sizeof(w), "barrier: fences, load poll, %d blocks", n2);
        if (time_graph([&](int) { k_bar<1><<<n2, 256, 0, s>>>(bar, part, out, n2); }, w)) return 1;
        snprintf(w, sizeof(w), "two kernels, %d blocks", n2);
        if (time_graph([&](int) { k_p1<<<n2, 256, 0, s>>>(part); k_p2<<<n2, 256, 0, s>>>(part, out, n2); }, w)) return 1;
    }
    return 0;
}
