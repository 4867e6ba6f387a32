// How long does a workgroup barrier take on gfx950, as a function of the waves in the workgroup and of what precedes it?
//   hipcc --offload-arch=gfx950 -O3 -o barrier_probe barrier_probe.hip && ./barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void k(long long* out, int iters) {
    __shared__ int s[64];
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) __builtin_amdgcn_s_barrier();
        if (MODE == 1) __syncthreads();
        if (MODE == 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        if (MODE == 3) { s[threadIdx.x & 63] = i; __syncthreads(); }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

int main() {
    long long* d; hipMalloc(&d, 8);
    const int iters = 2000;
    for (int waves : {1, 2, 4, 8, 12, 16}) {
        long long r[4];
        for (int m = 0; m < 4; ++m) {
            for (int rep = 0; rep < 2; ++rep) {
                if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, d, iters);
                if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, d, iters);
                if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), 0, 0, d, iters);
                if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(64 * waves), 0, 0, d, iters);
                hipDeviceSynchronize();
            }
            hipMemcpy(&r[m], d, 8, hipMemcpyDeviceToHost);
        }
        printf("%2d waves: s_barrier %.1f | __syncthreads %.1f | waitcnt+s_barrier %.1f | lds write + __syncthreads %.1f  (clock64 ticks per iteration)\n", waves,
               (double)r[0] / iters, (double)r[1] / iters, (double)r[2] / iters, (double)r[3] / iters);
    }
    return 0;
}
