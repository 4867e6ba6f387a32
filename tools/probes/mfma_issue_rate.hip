// mfma_issue_rate.hip — issue cost of the bf16 MFMA shapes on gfx950 (VERDICT r3 next #2a: is 16x16x16 half of 16x16x32, so that QK^T at
// d = 40 could run as one K32 + one K16 step — 48 instead of 64 padded columns?).  One wave per SIMD, 8 independent accumulators,
// 4096 MFMAs per wave, timed with s_memtime on wave 0 of every workgroup; 256 workgroups (one per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_issue_rate.hip -o tools/probes/mfma_issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void __launch_bounds__(256) probe(unsigned long long* out, float* sink, int iters) {
    bf8 a8, b8; s4 a4, b4;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(0.001f * (threadIdx.x & 7)); b8[i] = (__bf16)0.5f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { a4[i] = (short)(0x3c00 + threadIdx.x); b4[i] = 0x3f00; }
    f4 acc[8]; f16v big[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { f4 z = {0.f, 0.f, 0.f, 0.f}; acc[i] = z; }
#pragma unroll
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // the MFMAs are issued from inline asm on accumulators tied in place ("+v"): through the builtins hipcc rotates the loop-carried
    // accumulators through overlapping AGPR windows (a[24:27] = mfma(.., a[22:25])) or shuttles them VGPR <-> AGPR every iteration,
    // which chains consecutive MFMAs and triples the measured cost
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a8), "v"(b8));
            else if (KIND == 1) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a4), "v"(b4));
            else if (KIND == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[i & 3]) : "v"(a8), "v"(b8));
            else asm volatile("v_mfma_f32_32x32x8_bf16 %0, %1, %2, %0" : "+v"(big[i & 3]) : "v"(a4), "v"(b4));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += big[i][0];
    if (s == 12345.f) sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND> void run(const char* name, double flop_per_mfma) {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 4 * 8); hipMalloc(&sink, 1024);
    const int iters = 512;
    hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(256), 0, 0, out, sink, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(256), 0, 0, out, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(1024);
    hipMemcpy(h.data(), out, 1024 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double n = iters * 8.0;
    printf("%-28s %6.1f s_memtime ticks per MFMA (median wave), kernel %.1f us for %d MFMAs per wave = %.1f ns each = %.0f TFLOP/s chip\n",
           name, (double)h[512] / n, ms * 1e3, (int)n, ms * 1e6 / n, 256 * 4 * n * flop_per_mfma / (ms * 1e-3) / 1e12);
}

int main() {
    run<0>("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32);
    run<1>("v_mfma_f32_16x16x16_bf16", 2.0 * 16 * 16 * 16);
    run<2>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16);
    run<3>("v_mfma_f32_32x32x8_bf16", 2.0 * 32 * 32 * 8);
    return 0;
}
