// A __global__ TEMPLATE in a header (weak ODR linkage): every translation unit that includes it emits the kernel into its own code object
// AND a host stub of the same weak, hidden name; the linker keeps ONE stub while all eleven code objects register the kernel against it —
// the round-5 form of hcp_fill32_kernel (csrc/hcp_device.h at b39b1d4, before it became `static`).
#pragma once
#include <hip/hip_runtime.h>
#ifndef STUB_LINKAGE
#define STUB_LINKAGE            /* empty = weak ODR linkage (the suspected form); the fix compiles with -DSTUB_LINKAGE=static */
#endif
template <int UNUSED>
STUB_LINKAGE __global__ void __launch_bounds__(256) shared_fill_kernel(unsigned* p, unsigned v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
static inline int shared_fill(unsigned* p, unsigned v, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(shared_fill_kernel<0>, dim3(64), dim3(256), 0, s, p, v, n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
