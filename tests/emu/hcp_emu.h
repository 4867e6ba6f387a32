// hcp_emu.h — TEST INFRASTRUCTURE ONLY (never part of libhcp_mi355x.so).
//
// A minimal wave64 interpreter for the kernels in hcp_diffusion_amd/csrc: every GPU thread
// is a user-space fibre; __syncthreads() and the wave collectives (MFMA, shuffles) are
// rendezvous points resolved by a round-robin scheduler (hcp_emu.cpp).  MFMA fragment
// layouts follow cdna_hip_programming.md §3, so an indexing bug in a kernel shows up as a
// wrong answer on the CPU, before a GPU-minute is spent.  Built only by tests/emu/build_emu.py
// with host clang++ and -DHCP_EMU.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>

namespace hcp_emu {
struct uint3_t { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
enum { ST_RUN = 0, ST_BARRIER = 1, ST_WAVE = 2, ST_DONE = 3 };
enum { OP_SHFL = 1, OP_MFMA16 = 2, OP_MFMA32 = 3, OP_TR16 = 4 };
struct Fiber {
    void* sp;
    uint3_t tid;
    int lin, wave, lane, state;
    int op;
    uint32_t in[24];
    uint32_t out[16];
    int src_lane;
    char* stack;
    struct Pend { unsigned char* dst; unsigned char data[16]; int n; }* pend;      // LDS-DMA writes issued and not yet landed (deferred mode)
    int pend_n;
};
// LDS-DMA timing model.  0 (default): a copy lands the moment it is issued — the EARLIEST the hardware allows, which is what exposes
// write-after-read hazards (a refill overwriting fragments still being read).  1: a copy lands only when its issuing lane executes the
// wait that retires it (counted vmcnt: all but the N newest; HCP_SYNC / hcp_dma_wait_all: everything) — the LATEST the hardware allows,
// which is what exposes read-after-write hazards (a tile read before the wait + barrier that publish it).  A counted-vmcnt / raw-barrier
// protocol has to pass in both modes (tests/test_kernels.py::test_gemm_dma_protocols_under_late_landing).
extern int g_dma_deferred;
void dma_push(void* dst, const void* src, int n);       // src == nullptr: zeros (out-of-range lanes of a buffer load)
void dma_drain(int keep);
extern Fiber* g_cur;
extern uint3_t g_block;
extern dim3 g_bdim, g_gdim;
extern unsigned char* g_smem;
void yield_barrier();
void wave_collective();
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
}  // namespace hcp_emu

using hcp_emu::dim3;
typedef void* hipStream_t;

#define threadIdx (hcp_emu::g_cur->tid)
#define blockIdx (hcp_emu::g_block)
#define blockDim (hcp_emu::g_bdim)
#define gridDim (hcp_emu::g_gdim)

#define HCP_DEVICE static inline
#define HCP_MEMBER inline
#define HCP_KERNEL(maxthreads) static void
#define HCP_DYN_SMEM(name) unsigned char* name = hcp_emu::g_smem
#define HCP_SYNC() (hcp_emu::dma_drain(0), hcp_emu::yield_barrier())      // __syncthreads() with LDS-DMA in flight carries vmcnt(0)
#define HCP_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hcp_emu::launch(grid, block, smem, [=]() { kernel(__VA_ARGS__); })

typedef short hcp_bf16x8 __attribute__((ext_vector_type(8)));
typedef short hcp_bf16x4 __attribute__((ext_vector_type(4)));
typedef short hcp_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hcp_f32x4 __attribute__((ext_vector_type(4)));
typedef float hcp_f32x16 __attribute__((ext_vector_type(16)));

HCP_DEVICE hcp_f32x4 hcp_mfma16(hcp_bf16x8 a, hcp_bf16x8 b, hcp_f32x4 c) {
    hcp_emu::Fiber* f = hcp_emu::g_cur;
    f->op = hcp_emu::OP_MFMA16;
    memcpy(&f->in[0], &a, 16); memcpy(&f->in[4], &b, 16); memcpy(&f->in[8], &c, 16);
    hcp_emu::wave_collective();
    hcp_f32x4 d; memcpy(&d, hcp_emu::g_cur->out, 16); return d;
}
HCP_DEVICE hcp_f32x16 hcp_mfma32(hcp_bf16x8 a, hcp_bf16x8 b, hcp_f32x16 c) {
    hcp_emu::Fiber* f = hcp_emu::g_cur;
    f->op = hcp_emu::OP_MFMA32;
    memcpy(&f->in[0], &a, 16); memcpy(&f->in[4], &b, 16); memcpy(&f->in[8], &c, 64);
    hcp_emu::wave_collective();
    hcp_f32x16 d; memcpy(&d, hcp_emu::g_cur->out, 64); return d;
}
HCP_DEVICE uint32_t hcp_emu_shfl_u32(uint32_t v, int src) {
    hcp_emu::Fiber* f = hcp_emu::g_cur;
    f->op = hcp_emu::OP_SHFL; f->in[0] = v; f->src_lane = src & 63;
    hcp_emu::wave_collective();
    return hcp_emu::g_cur->out[0];
}
HCP_DEVICE float hcp_shfl(float v, int src) {
    uint32_t u; memcpy(&u, &v, 4); u = hcp_emu_shfl_u32(u, src); memcpy(&v, &u, 4); return v;
}
HCP_DEVICE float hcp_shfl_xor(float v, int mask) { return hcp_shfl(v, hcp_emu::g_cur->lane ^ mask); }
HCP_DEVICE int hcp_shfl_xor_i(int v, int mask) {
    return (int)hcp_emu_shfl_u32((uint32_t)v, hcp_emu::g_cur->lane ^ mask);
}
HCP_DEVICE void hcp_atomic_add(float* p, float v) { *p += v; }
HCP_DEVICE int hcp_lane() { return hcp_emu::g_cur->lane; }
HCP_DEVICE float hcp_exp2(float x) { return exp2f(x); }
// LDS transpose read (ds_read_b64_tr_b16), semantics as measured on gfx950 (tools/probes/tr_read_probe.hip)
HCP_DEVICE hcp_bf16x4 hcp_lds_read_tr4(const unsigned short* p) {
    hcp_emu::Fiber* f = hcp_emu::g_cur;
    f->op = hcp_emu::OP_TR16;
    memcpy(&f->in[0], p, 8);
    hcp_emu::wave_collective();
    hcp_bf16x4 d; memcpy(&d, hcp_emu::g_cur->out, 8); return d;
}
// LDS-DMA (global_load_lds_dwordx4): lane l's 16 bytes land at wave_base + 16*l; the interpreter copies immediately.
HCP_DEVICE void hcp_glds16(const void* gsrc, void* lds_wave_base) {
    hcp_emu::dma_push((unsigned char*)lds_wave_base + 16 * hcp_emu::g_cur->lane, gsrc, 16);
}
struct hcp_rsrc { const unsigned char* base; unsigned nbytes; };
#define HCP_BUF_OOB 0x80000000u
HCP_DEVICE hcp_rsrc hcp_make_rsrc(const void* base) { hcp_rsrc r; r.base = (const unsigned char*)base; r.nbytes = 0x7fffffffu; return r; }
HCP_DEVICE hcp_rsrc hcp_make_rsrc_n(const void* base, unsigned nbytes) { hcp_rsrc r; r.base = (const unsigned char*)base; r.nbytes = nbytes; return r; }
HCP_DEVICE void hcp_buf_glds16(hcp_rsrc rsrc, unsigned voffset, void* lds_wave_base) {
    unsigned char* dst = (unsigned char*)lds_wave_base + 16 * hcp_emu::g_cur->lane;
    hcp_emu::dma_push(dst, (voffset >= rsrc.nbytes || voffset + 16 > rsrc.nbytes) ? nullptr : rsrc.base + voffset, 16);
}
HCP_DEVICE hcp_bf16x8 hcp_buf_load16(hcp_rsrc rsrc, unsigned voffset) {
    hcp_bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (voffset < rsrc.nbytes && voffset + 16 <= rsrc.nbytes) memcpy(&v, rsrc.base + voffset, 16);
    return v;
}
HCP_DEVICE hcp_bf16x4 hcp_buf_load8(hcp_rsrc rsrc, unsigned voffset) {
    hcp_bf16x4 v = {0, 0, 0, 0};
    if (voffset < rsrc.nbytes && voffset + 8 <= rsrc.nbytes) memcpy(&v, rsrc.base + voffset, 8);
    return v;
}
HCP_DEVICE hcp_f32x4 hcp_buf_load16f(hcp_rsrc rsrc, unsigned voffset) {
    hcp_f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (voffset < rsrc.nbytes && voffset + 16 <= rsrc.nbytes) memcpy(&v, rsrc.base + voffset, 16);
    return v;
}
struct hcp_desc4 { const unsigned char* base; unsigned nbytes; };
HCP_DEVICE hcp_desc4 hcp_make_desc(const void* base, unsigned nbytes) { hcp_desc4 d; d.base = (const unsigned char*)base; d.nbytes = nbytes; return d; }
HCP_DEVICE void hcp_dma16(hcp_desc4 d, unsigned voffset, void* lds_wave_base) {
    unsigned char* dst = (unsigned char*)lds_wave_base + 16 * hcp_emu::g_cur->lane;
    hcp_emu::dma_push(dst, (voffset >= d.nbytes || voffset + 16 > d.nbytes) ? nullptr : d.base + voffset, 16);
}
template <int P> HCP_DEVICE void hcp_setprio() {}
HCP_DEVICE void hcp_sched_fence() {}
#define HCP_IS_EMULATED 1
#define HCP_LAUNCH_CHECK(name) return 0
static inline int hcp_memset_async(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline int hcp_memcpy_async(void* d, const void* s, size_t n, hipStream_t) { memmove(d, s, n); return 0; }
#define HCP_WAVES_PER_SIMD(n)
template <int MASK, int N> HCP_DEVICE void hcp_sched_group() {}
HCP_DEVICE float hcp_max16(const hcp_f32x4 (&s)[4]) {
    float m = s[0][0];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 4; ++r) m = fmaxf(m, s[k][r]);
    return m;
}
HCP_DEVICE float hcp_max8(const hcp_f32x4 (&s)[2]) {
    float m = s[0][0];
    for (int k = 0; k < 2; ++k) for (int r = 0; r < 4; ++r) m = fmaxf(m, s[k][r]);
    return m;
}
HCP_DEVICE void hcp_dma_wait_all() { hcp_emu::dma_drain(0); }
HCP_DEVICE int hcp_uniform(int v) { return v; }
HCP_DEVICE void hcp_force_ready(hcp_bf16x8&) {}
HCP_DEVICE void hcp_force_ready(float&) {}
HCP_DEVICE void hcp_force_ready(int&) {}
#define HCP_DEVICE_GLOBAL static
HCP_DEVICE void hcp_wait_vmcnt(int n) { hcp_emu::dma_drain(n); }
HCP_DEVICE void hcp_barrier_keep_dma() { hcp_emu::yield_barrier(); }
HCP_DEVICE void hcp_barrier_only() { hcp_emu::yield_barrier(); }
template <int N> HCP_DEVICE void hcp_wait_vmcnt_c() { hcp_emu::dma_drain(N); }
HCP_DEVICE bool hcp_all(bool pred) {
    int v = pred ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) v &= hcp_shfl_xor_i(v, m);
    return v != 0;
}
