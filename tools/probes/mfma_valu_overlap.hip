// mfma_valu_overlap.hip — does VALU / transcendental work of one wave overlap with the MFMAs of another wave ON THE SAME SIMD (gfx950)?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o tools/probes/mfma_valu_overlap
// One 512-thread workgroup per CU: waves 0-3 land on SIMDs 0-3, waves 4-7 on the same SIMDs again.  MODE bits: 1 = waves 0-3 run an
// MFMA chain, 2 = waves 4-7 run a VALU loop (KIND 0: v_exp_f32, 1: v_mul_f32, 2: v_cvt_pk_bf16_f32 + v_max3 mix), 4 = ONE wave group
// runs both interleaved in one instruction stream (waves 4-7 idle).  Time per iteration tells whether the pipes add or overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void valu_block(float (&x)[16]) {
    if (KIND == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
    } else if (KIND == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = x[i] * 1.0001f;
    } else {
#pragma unroll
        for (int i = 0; i < 16; i += 2) { float m = fmaxf(fmaxf(x[i], x[i + 1]), 0.5f); x[i] = m * 0.999f; x[i + 1] += 1e-3f; }
    }
}

template <int MODE, int KIND>
__global__ void __launch_bounds__(512) probe(float* out, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { f4 z = {0.f, 0.f, 0.f, 0.f}; acc[i] = z; }
    bf8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x & 7)); b[i] = (__bf16)0.5f; }
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * threadIdx.x + i;
    if (MODE & 4) {
        if (wave < 4)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {              // 8 MFMAs and 16 VALU ops of kind KIND interleaved 1 : 2
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
                    if (KIND == 0) { x[2 * i] = __builtin_amdgcn_exp2f(x[2 * i]); x[2 * i + 1] = __builtin_amdgcn_exp2f(x[2 * i + 1]); }
                    else { x[2 * i] *= 1.0001f; x[2 * i + 1] *= 1.0001f; }
                }
            }
    } else {
        if ((MODE & 1) && wave < 4)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
            }
        if ((MODE & 2) && wave >= 4)
            for (int it = 0; it < iters; ++it) valu_block<KIND>(x);
        if ((MODE & 8))                                  // every wave runs the VALU loop: 2 waves per SIMD
            for (int it = 0; it < iters; ++it) valu_block<KIND>(x);
        if ((MODE & 16))                                 // every wave runs the MFMA loop: 2 waves per SIMD
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
            }
        if ((MODE & 32))                                 // every wave runs the interleaved stream: 2 waves per SIMD
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
                    if (KIND == 0) { x[2 * i] = __builtin_amdgcn_exp2f(x[2 * i]); x[2 * i + 1] = __builtin_amdgcn_exp2f(x[2 * i + 1]); }
                    else { x[2 * i] *= 1.0001f; x[2 * i + 1] *= 1.0001f; }
                }
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int KIND>
static float run(float* o, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE, KIND>), dim3(256), dim3(512), 0, 0, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    return best * 1e6f / iters;          // ns per iteration (8 MFMAs and / or 16 VALU ops per wave)
}

int main() {
    float* o; hipMalloc(&o, 256 * 512 * 4);
    const int iters = 20000;
    const char* kn[3] = {"v_exp_f32", "v_mul_f32", "max/mul/add mix"};
    printf("per iteration: 8 x mfma_16x16x32_bf16 (waves 0-3) and / or 16 VALU ops (waves 4-7, same SIMDs); ns, and cycles at 2.4 GHz\n");
    float m = run<1, 0>(o, iters);
    printf("MFMA alone                    %7.1f ns (%5.1f cyc / MFMA)\n", m, m * 2.4f / 8);
    float v0 = run<2, 0>(o, iters), v1 = run<2, 1>(o, iters), v2 = run<2, 2>(o, iters);
    printf("VALU alone   %-16s %7.1f ns (%5.1f cyc / op)\n", kn[0], v0, v0 * 2.4f / 16);
    printf("VALU alone   %-16s %7.1f ns (%5.1f cyc / op)\n", kn[1], v1, v1 * 2.4f / 16);
    printf("VALU alone   %-16s %7.1f ns (%5.1f cyc / op)\n", kn[2], v2, v2 * 2.4f / 16);
    float b0 = run<3, 0>(o, iters), b1 = run<3, 1>(o, iters), b2 = run<3, 2>(o, iters);
    printf("two waves    MFMA || %-10s %7.1f ns  (sum %.1f, max %.1f)\n", kn[0], b0, m + v0, std::max(m, v0));
    printf("two waves    MFMA || %-10s %7.1f ns  (sum %.1f, max %.1f)\n", kn[1], b1, m + v1, std::max(m, v1));
    printf("two waves    MFMA || %-10s %7.1f ns  (sum %.1f, max %.1f)\n", kn[2], b2, m + v2, std::max(m, v2));
    float w0 = run<8, 0>(o, iters), w1 = run<8, 1>(o, iters), w2 = run<8, 2>(o, iters), wm = run<16, 0>(o, iters);
    printf("2 waves/SIMD all VALU  %-16s %7.1f ns per iteration of BOTH waves (one wave alone %.1f)\n", kn[0], w0, v0);
    printf("2 waves/SIMD all VALU  %-16s %7.1f ns (one wave alone %.1f)\n", kn[1], w1, v1);
    printf("2 waves/SIMD all VALU  %-16s %7.1f ns (one wave alone %.1f)\n", kn[2], w2, v2);
    printf("2 waves/SIMD all MFMA                   %7.1f ns (one wave alone %.1f)\n", wm, m);
    float j0 = run<32, 0>(o, iters), j1 = run<32, 1>(o, iters);
    printf("2 waves/SIMD interleaved MFMA + 2 x %-9s %7.1f ns\n", kn[0], j0);
    printf("2 waves/SIMD interleaved MFMA + 2 x %-9s %7.1f ns\n", kn[1], j1);
    float i0 = run<4, 0>(o, iters), i1 = run<4, 1>(o, iters);
    printf("one wave     MFMA, 2 x %-8s interleaved %7.1f ns  (sum %.1f)\n", kn[0], i0, m + v0);
    printf("one wave     MFMA, 2 x %-8s interleaved %7.1f ns  (sum %.1f)\n", kn[1], i1, m + v1);
    hipFree(o);
    return 0;
}
