// launch_floor.hip -- what a kernel node costs inside a replayed HIP graph on this GPU: empty, and with the memory
// behaviours of the PSD step's small kernels (read N bytes, write N bytes plain / nontemporal), chained by dependencies.
// build: hipcc --offload-arch=gfx950 -O2 tools/exp/launch_floor.hip -o tools/exp/launch_floor ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_empty(int *p) {
    extern __shared__ int lds[];
    if (p && threadIdx.x == 12345) p[0] = lds[0];
}
// mode 0: read n uint4, 1: write plain, 2: write nontemporal, 3: read + write plain (elementwise), 4: read + write nt
__global__ void k_mem(const uint4 *__restrict__ in, uint4 *__restrict__ out, long long n, int mode) {
    uint4 acc = {0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        uint4 v = {1, 2, 3, 4};
        if (mode == 0 || mode >= 3) v = in[i];
        if (mode == 0) { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
        if (mode == 1 || mode == 3) out[i] = v;
        if (mode == 2 || mode == 4) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 w = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(out + i));
        }
    }
    if (mode == 0 && acc.x == 0x12345678u) out[0] = acc;
}
int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    CK(hipFuncSetAttribute((const void *)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
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
        printf("%-60s %6.2f us per kernel node\n", what, ms * 1e3 / (reps * nodes));
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
        return 0;
    };
    if (time_graph([&](int) { k_empty<<<1, 64, 0, s>>>(nullptr); }, "empty, 1 x 64")) return 1;
    if (time_graph([&](int) { k_empty<<<232, 768, 96 * 1024, s>>>(nullptr); }, "empty, 232 x 768, 96 KB LDS")) return 1;
    uint4 *A, *B;
    const long long maxb = 64ll << 20;
    CK(hipMalloc((void **)&A, maxb));
    CK(hipMalloc((void **)&B, maxb));
    CK(hipMemset(A, 1, maxb));
    const char *names[5] = {"read", "write", "write nontemporal", "read + write", "read + write nontemporal"};
    for (long long bytes : {64ll << 10, 1ll << 20, 5632ll << 10, 16ll << 20}) {
        for (int mode = 0; mode < 5; ++mode) {
            char what[128];
            snprintf(what, sizeof(what), "%s %lld KB (ping-pong buffers, 1024 x 256)", names[mode], bytes >> 10);
            const long long n = bytes / 16;
            if (time_graph([&](int i) { k_mem<<<1024, 256, 0, s>>>((i & 1) ? A : B, (i & 1) ? B : A, n, mode); }, what)) return 1;
        }
    }
    return 0;
}
