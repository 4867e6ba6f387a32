// tools/probes/launch_gap_probe.cpp — what does ONE dependent kernel boundary cost on this stack, by submission path?
// N back-to-back launches of (a) an empty kernel, (b) a kernel shaped like the GEMM family (256 workgroups x 768 threads, 150 KB of
// dynamic LDS, one global store per wave), (c) a small streaming kernel (8 MB copy), submitted
//   1. one by one into a stream (hipLaunchKernelGGL from C++),
//   2. as a captured hipGraph replayed with hipGraphLaunch (what torch.cuda.CUDAGraph does),
// timed with HIP events around the whole batch: us per launch = floor of a step that is a chain of ~900 dependent launches.
//   build: hipcc -O2 --offload-arch=gfx950 tools/probes/launch_gap_probe.cpp -o tools/probes/launch_gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_empty() {}
__global__ void __launch_bounds__(768) k_gemm_shaped(float* out) {
    extern __shared__ unsigned char smem[];
    if ((threadIdx.x & 63) == 0) { smem[threadIdx.x] = 1; out[blockIdx.x * 12 + (threadIdx.x >> 6)] = 1.0f; }
}
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) b[i] = a[i];
}

template <typename F>
static void run(const char* name, F launch, hipStream_t st, int N) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipStreamSynchronize(st));
    float best_s = 1e9f, best_g = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < N; ++i) launch();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best_s) best_s = ms;
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) launch();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best_g) best_g = ms;
    }
    printf("%-44s stream %6.2f us/launch   hipGraph %6.2f us/launch\n", name, best_s * 1e3f / N, best_g * 1e3f / N);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const int N = 1000;
    float* out; CK(hipMalloc(&out, 1 << 20));
    float4 *a, *b; CK(hipMalloc(&a, 8 << 20)); CK(hipMalloc(&b, 8 << 20));
    CK(hipFuncSetAttribute((const void*)k_gemm_shaped, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    run("empty kernel, 1 x 64", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); }, st, N);
    run("empty kernel, 256 x 256", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st); }, st, N);
    run("empty kernel, 2048 x 256", [&] { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, st); }, st, N);
    run("GEMM-shaped, 256 x 768, 150 KB LDS", [&] { hipLaunchKernelGGL(k_gemm_shaped, dim3(256), dim3(768), 150 * 1024, st, out); }, st, N);
    run("GEMM-shaped, 512 x 768, 150 KB LDS", [&] { hipLaunchKernelGGL(k_gemm_shaped, dim3(512), dim3(768), 150 * 1024, st, out); }, st, N);
    run("GEMM-shaped, 256 x 768, 40 KB LDS", [&] { hipLaunchKernelGGL(k_gemm_shaped, dim3(256), dim3(768), 40 * 1024, st, out); }, st, N);
    run("copy 8 MB, 2048 x 256", [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, st, a, b, (8 << 20) / 16); }, st, N);
    run("copy 1 MB, 256 x 256", [&] { hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, st, a, b, (1 << 20) / 16); }, st, N);
    return 0;
}
