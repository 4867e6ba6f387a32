// GPU probe: semantics of buffer_load_dwordx4 ... offen lds (raw_ptr_buffer_load_lds) on gfx950:
//   (1) lane l's 16 bytes land at lds_base + 16*l;  (2) a lane with voffset >= num_records writes ZEROS;
//   (3) rebasing the resource by a negative element offset works (the base pointer is what moves, not soffset).
// hipcc --offload-arch=gfx950 buffer_lds_probe.hip -o /tmp/blp && /tmp/blp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* src, unsigned* out) {
    extern __shared__ unsigned char smem[];
    for (int i = threadIdx.x; i < 1024; i += 64) ((unsigned*)smem)[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(src + 256 - 64), (short)0, 0x7fffffff, 0x00020000);   // base moved back by 64 dwords
    unsigned voff = threadIdx.x * 16 + 64 * 4;            // so lane l reads src[256 + 4l .. +3]
    if (threadIdx.x % 3 == 0) voff = 0x80000000u;          // masked lanes
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, (int)voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    unsigned *d, *o;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 256 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o);
    std::vector<unsigned> r(256);
    hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            unsigned want = (l % 3 == 0) ? 0u : 256u + 4 * l + j;
            if (r[4 * l + j] != want) { if (bad < 8) printf("lane %d word %d: got %u want %u\n", l, j, r[4 * l + j], want); ++bad; }
        }
    printf(bad ? "FAIL (%d mismatches)\n" : "buffer->LDS semantics OK (lane-linear, OOB lanes write zeros, rebased resource)\n", bad);
    return bad != 0;
}
