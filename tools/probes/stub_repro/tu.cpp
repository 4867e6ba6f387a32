// One of N translation units of the reproducer: clears an accumulator through the header's fill kernel, then adds into it with fp32
// atomics from 2048 workgroups (the shape of lora_wgrad / the query-split dK/dV of round 5).  -DTU=<n> names the entry point.
#include "shared_kernel.h"
#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)
namespace {
__global__ void __launch_bounds__(256) CAT(add_kernel_, TU)(float* acc, int n) {
    atomicAdd(acc + (blockIdx.x * 37 + threadIdx.x * 101) % n, (float)((threadIdx.x & 3) + 1));
}
}
extern "C" __attribute__((visibility("default"))) int CAT(run_tu_, TU)(float* acc, int n, int use_fill, hipStream_t s) {
    if (use_fill && shared_fill((unsigned*)acc, 0u, (size_t)n, s)) return -1;
    hipLaunchKernelGGL(CAT(add_kernel_, TU), dim3(2048), dim3(256), 0, s, acc, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
