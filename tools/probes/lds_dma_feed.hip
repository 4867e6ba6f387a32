// GPU probe: how fast can ONE CU pull L2 / MALL-resident data into LDS with buffer_load_dwordx4 ... lds (1 KiB per wave-instruction),
// as a function of the number of loader waves and of the pieces each wave keeps in flight?  (is the GEMM main loop's
// 46-54 GB/s per CU a limit of the path or of the bytes in flight?)
// hipcc -O3 --offload-arch=gfx950 lds_dma_feed.hip -o lds_dma_feed && ./lds_dma_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int D>   // pieces in flight per wave
__global__ void feed(const unsigned char* src, size_t region_bytes, int regions, int iters, unsigned* sink) {
    extern __shared__ unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const unsigned char* base = src + (size_t)(blockIdx.x % regions) * region_bytes;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, 0x7fffffff, 0x00020000);
    unsigned char* ring = smem + (size_t)wave * D * 1024;
    const unsigned span = (unsigned)region_bytes;
    unsigned off = (unsigned)(wave * 1024 + lane * 16);
#pragma unroll 1
    for (int it = 0; it < iters; it += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(ring + j * 1024), 16, (int)off, 0, 0, 0);
            off += (unsigned)nw * 1024;
            if (off >= span) off -= span;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");       // at most D - 1 older pieces still outstanding
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = ((unsigned*)smem)[lane];
}

template <int D>
static void run(const unsigned char* src, size_t region, int regions, int nw, int grid, unsigned* sink, const char* what) {
    const int iters = 4096 / D * D;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (size_t)nw * D * 1024;
    hipFuncSetAttribute((const void*)feed<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(feed<D>, dim3(grid), dim3(64 * nw), lds, 0, src, region, regions, iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * nw * iters * 1024.0;
    printf("%-18s waves %2d  in flight/wave %2d KB (CU %3d KB)  %7.1f GB/s per CU  %6.2f TB/s chip  (%.3f ms)\n", what, nw, D, nw * D,
           bytes / grid / (ms * 1e-3) / 1e9 * (grid > 256 ? grid / 256.0 : 1.0), bytes / (ms * 1e-3) / 1e12, ms);
}

int main(int argc, char**) {
    const bool quick = argc > 1;
    const size_t total = 512ull << 20;
    unsigned char* src; unsigned* sink;
    hipMalloc(&src, total); hipMemset(src, 1, total); hipMalloc(&sink, 4096 * 4);
    struct { size_t region; int regions; const char* what; } cases[] = {
        {256 << 10, 1, "shared 256 KB (L2)"}, {16 << 10, 256, "own 16 KB (L2)"}, {64 << 10, 256, "own 64 KB (L2)"}, {112 << 10, 256, "own 112 KB (L2)"},
        {256 << 10, 256, "own 256 KB (MALL)"}, {2 << 20, 256, "own 2 MB (MALL/HBM)"}};
    for (auto& c : cases)
        for (int nw : {1, 2, 4, 8}) {
            if (quick && nw != 4) continue;
            if (!quick) { run<2>(src, c.region, c.regions, nw, 256, sink, c.what); run<4>(src, c.region, c.regions, nw, 256, sink, c.what); }
            run<9>(src, c.region, c.regions, nw, 256, sink, c.what);
            run<18>(src, c.region, c.regions, nw, 256, sink, c.what);
            if (nw <= 4 && !quick) run<36>(src, c.region, c.regions, nw, 256, sink, c.what);
        }
    return 0;
}
