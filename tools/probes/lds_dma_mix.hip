// GPU probe: do LDS-DMA landings (buffer_load_dwordx4 ... lds), ds_read_b128 fragment reads and MFMAs of OTHER waves of the same CU
// overlap, or do they take turns?  One workgroup per CU: NL loader waves stream L2-hot data into an LDS ring, NC consumer waves
// read another LDS region (ds_read_b128, conflict-free) and / or issue MFMAs; each role runs a fixed amount of work, the kernel ends
// when all are done.  Compare t(both) with t(dma) and t(consume).
// hipcc -O3 --offload-arch=gfx950 lds_dma_mix.hip -o lds_dma_mix && ./lds_dma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// mode bits: 1 = loaders run, 2 = consumers read LDS, 4 = consumers issue MFMAs (on the values read when bit 1, else on registers)
__global__ void mix(const unsigned char* src, int nl, int mode, int dma_pieces, int reads, int mfmas_per_read, unsigned* sink, unsigned* role_ticks) {
    extern __shared__ unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned acc0 = 0;
    const unsigned long long t_begin = wall_clock64();
    if (wave < nl) {
        if (mode & 1) {
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, 0x7fffffff, 0x00020000);
            unsigned char* ring = smem + 65536 + (size_t)wave * 9 * 1024;
            unsigned off = (unsigned)(wave * 1024 + lane * 16);
#pragma unroll 1
            for (int it = 0; it < dma_pieces; it += 9) {
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(ring + j * 1024), 16, (int)off, 0, 0, 0);
                    off += (unsigned)nl * 1024;
                    if (off >= (256u << 10)) off -= (256u << 10);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else if (mode & 6) {
        const int cw = wave - nl;
        const unsigned char* base = smem + (cw & 3) * 16384 + lane * 16;      // 64 lanes x 16 B contiguous = conflict-free b128
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll 1
        for (int it = 0; it < reads; it += 4) {
            bf16x8 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (mode & 2) v[j] = *(const bf16x8*)(base + ((it + j) & 15) * 1024);
                else v[j] = a;
            }
            if (mode & 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    for (int m = 0; m < mfmas_per_read; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v[j], b, acc, 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc0 += (unsigned)v[j][0];
            }
        }
        acc0 += (unsigned)acc[0];
    }
    const unsigned long long t_end = wall_clock64();
    __syncthreads();
    if (lane == 0) { sink[blockIdx.x * 16 + wave] = acc0; role_ticks[blockIdx.x * 16 + wave] = (unsigned)(t_end - t_begin); }
}

static unsigned* g_ticks; static float g_loader_us, g_consumer_us;
static float run(const unsigned char* src, int nl, int nc, int mode, int dma_pieces, int reads, int mpr, unsigned* sink) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t lds = 65536 + (size_t)nl * 9 * 1024;
    (void)hipFuncSetAttribute((const void*)mix, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(mix, dim3(256), dim3(64 * (nl + nc)), lds, 0, src, nl, mode, dma_pieces, reads, mpr, sink, g_ticks);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    static unsigned h[256 * 16];
    (void)hipMemcpy(h, g_ticks, sizeof(h), hipMemcpyDeviceToHost);
    double lo = 0, co = 0;                                   // mean over workgroups of the slowest wave of each role (100 MHz wall clock)
    for (int b = 0; b < 256; ++b) {
        unsigned ml = 0, mc = 0;
        for (int w = 0; w < nl + nc; ++w) { unsigned v = h[b * 16 + w]; if (w < nl) ml = v > ml ? v : ml; else mc = v > mc ? v : mc; }
        lo += ml; co += mc;
    }
    g_loader_us = (float)(lo / 256 / 100.0); g_consumer_us = (float)(co / 256 / 100.0);
    return ms * 1e3f;
}

int main() {
    unsigned char* src; unsigned* sink;
    (void)hipMalloc(&src, 1 << 20); (void)hipMemset(src, 1, 1 << 20); (void)hipMalloc(&sink, 256 * 16 * 4); (void)hipMalloc(&g_ticks, 256 * 16 * 4);
    const int nl = 4, nc = 8;
    const int pieces = 9 * 400;                    // per loader wave: 3.6 MB per wave, 14.4 MB per CU
    const int reads = 4 * 1400;                    // per consumer wave: 5600 ds_read_b128 = 5.6 MB per wave, 45 MB per CU
    printf("4 loader waves x %d KB, 8 consumer waves x %d ds_read_b128 (per CU), L2-hot source; times in us\n", pieces, reads);
    float t_dma = run(src, nl, nc, 1, pieces, reads, 1, sink);
    float t_rd = run(src, nl, nc, 2, pieces, reads, 1, sink);
    float t_both = run(src, nl, nc, 3, pieces, reads, 1, sink);
    printf("DMA only            %8.1f  (%.0f GB/s per CU)\n", t_dma, nl * pieces * 1024.0 / t_dma / 1e3);
    printf("LDS reads only      %8.1f  (%.0f B/clk per CU at 2.4 GHz)\n", t_rd, nc * reads * 1024.0 / (t_rd * 2400.0));
    printf("DMA + reads         %8.1f  (sum %.1f, max %.1f)   loaders done after %.1f us, readers after %.1f us\n", t_both, t_dma + t_rd, t_dma > t_rd ? t_dma : t_rd, g_loader_us, g_consumer_us);
    for (int mpr : {1, 2}) {
        float t_m = run(src, nl, nc, 4, pieces, reads, mpr, sink);
        float t_rm = run(src, nl, nc, 6, pieces, reads, mpr, sink);
        float t_dm = run(src, nl, nc, 5, pieces, reads, mpr, sink);
        float t_all = run(src, nl, nc, 7, pieces, reads, mpr, sink);
        printf("MFMA only (%d per read slot) %8.1f | reads+MFMA %8.1f | DMA+MFMA %8.1f | DMA+reads+MFMA %8.1f (loaders done after %.1f us)\n", mpr, t_m, t_rm, t_dm, t_all, g_loader_us);
    }
    return 0;
}
