// atomic_dq_probe.hip — what the dQ accumulation of a ONE-kernel attention backward costs on gfx950.
// The single-pass kernel owns 128 keys per workgroup and walks the 64-query tiles; every tile ends in a [64 q][40 d] fp32 partial of
// dQ that has to be ADDED into the (batch, head)'s dQ image by 32 workgroups (N = 4096): 32 x 21 MB = 671 MB of fp32 atomics per
// launch at B4 H8.  This probe issues exactly that traffic (same grid, same XCD-aware order, same MFMA accumulator lane layout:
// lane (fr, fg) owns dQ[q = 4 fg + r][d = 16 dt + fr]) and nothing else, as
//   0: global_atomic_add_f32, no return                     1: plain dword stores in the same pattern (the floor of the pattern)
//   2: the same atomics, image [bh][q][64] (256-byte rows)   3: atomics from a TRANSPOSED accumulator (lane owns 4 consecutive d of one q)
// so the kernel design can be priced before it is written.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/atomic_dq_probe.hip -o tools/probes/atomic_dq_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ int xcd_item(int id, int n) {
    const int q = n >> 3, r = n & 7, xcd = id & 7, slot = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

template <int KIND>
__global__ void __launch_bounds__(256) probe(float* img, int N, int nkb, int row, int spin) {
    const int item = xcd_item(blockIdx.x, gridDim.x);
    const int bh = item / nkb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
    float* base = img + (size_t)bh * N * row;
    float acc = 1.0f;
    for (int q0 = 0; q0 < N; q0 += 64) {
        for (int s = 0; s < spin; ++s) acc = acc * 1.0001f + 0.5f;     // stands in for the tile's MFMA time (dependent chain)
        float* t = base + (size_t)(q0 + 16 * wave) * row;
        if (KIND == 3) {
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = 16 * dt + 4 * fg + r;
                    if (d < 40) atomicAdd(t + fr * row + d, acc);
                }
        } else {
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = 16 * dt + fr;
                    float* p = t + (4 * fg + r) * row + d;
                    if (d < 40) {
                        if (KIND == 1) *p = acc; else atomicAdd(p, acc);
                    }
                }
        }
    }
}

template <int KIND> void run(const char* name, float* img, int B, int H, int N, int row, int spin) {
    const int nkb = N / 128, n = B * H * nkb;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipMemsetAsync(img, 0, (size_t)B * H * N * row * 4, 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<KIND>, dim3(n), dim3(256), 0, 0, img, N, nkb, row, spin);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double adds = (double)B * H * nkb * (N / 64) * 64 * 40;
    printf("%-44s row %3d spin %5d: %8.1f us   %.1f M element-adds  -> %.0f G adds/s  (%.2f TB/s of fp32)\n", name, row, spin, best * 1e3,
           adds / 1e6, adds / best / 1e6, adds * 4 / best / 1e9);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int B = 4, H = 8, N = 4096;
    float* img; hipMalloc(&img, (size_t)B * H * N * 64 * 4);
    for (int spin : {0, 400, 1600}) {
        run<0>("atomic f32, lane = d (64-B runs)", img, B, H, N, 40, spin);
        run<1>("plain dword stores, same pattern", img, B, H, N, 40, spin);
        run<2>("atomic f32, lane = d, 256-B rows", img, B, H, N, 64, spin);
        run<3>("atomic f32, lane = q (transposed accumulator)", img, B, H, N, 40, spin);
    }
    // half the partials: 256 keys per workgroup (nkb = N / 256) == the same kernel over N/2 key blocks
    return 0;
}
