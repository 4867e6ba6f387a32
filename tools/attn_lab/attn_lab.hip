// attn_lab.hip — GPU-only laboratory for the attention kernels (tools; never part of libhcp_mi355x.so).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I hcp_diffusion_amd/csrc tools/attn_lab/attn_lab.hip \
//         -L hcp_diffusion_amd -lhcp_mi355x -Wl,-rpath,'$ORIGIN/../../hcp_diffusion_amd' -o tools/attn_lab/attn_lab
//
// 1. probes the two hardware facts the DMA kernels lean on (exec-masked LDS-DMA lanes leave their LDS bytes untouched;
//    issue rate of the legacy 16x16x16 bf16 MFMA vs the gfx950 16x16x32 one);
// 2. times template variants of attn_dma.h against the first-generation product kernel (C ABI of the shipped library) on
//    the same random data, interleaved in one process (guide §5.4 rules 24/25), and checks every variant's O / lse
//    against it.
#include "attn_dma.h"
#include "attn_sw.h"
#include "attn_pp.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

using namespace hcp_attn;

extern "C" int hcp_attention_fwd(const void* Q, const void* K, const void* V, void* O, float* lse, int B, int H, int Nq, int Nk,
                                 int D, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs,
                                 float scale, const float* key_bias, long key_bias_bs, int causal, hipStream_t stream);
extern "C" int hcp_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                                 float* delta, void* dQ, void* dK, void* dV, int B, int H, int Nq, int Nk, int D, long q_bs,
                                 int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, float scale,
                                 const float* key_bias, long key_bias_bs, int causal, void* workspace, size_t workspace_bytes,
                                 hipStream_t stream);
extern "C" const char* hcp_last_error(void);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

// ---------------------------------------------------------------------------------------------- probe 1: masked DMA lanes
__global__ void probe_masked_dma(const unsigned* src, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 256; i += 64) ((unsigned*)smem)[i] = 0xdead0000u + i;
    __syncthreads();
    const hcp_desc4 r = hcp_make_desc(src, 60 * 16);                    // lanes 60..63 are out of range: zeros
    const unsigned voff = threadIdx.x * 16;
    if (threadIdx.x % 6 != 5) hcp_dma16(r, voff, smem);                 // every sixth lane masked off
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((unsigned*)smem)[i];
}
static bool run_probe_masked() {
    std::vector<unsigned> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
    unsigned *d, *o;
    CK(hipMalloc(&d, 1024)); CK(hipMalloc(&o, 1024));
    CK(hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe_masked_dma, dim3(1), dim3(64), 1024, 0, d, o);
    std::vector<unsigned> r(256);
    CK(hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            const unsigned want = (l % 6 == 5) ? 0xdead0000u + 4 * l + j : l >= 60 ? 0u : 1000u + 4 * l + j;
            if (r[4 * l + j] != want) { if (bad < 6) printf("  lane %d word %d: got %08x want %08x\n", l, j, r[4 * l + j], want); ++bad; }
        }
    printf("[probe] exec-masked LDS-DMA lanes keep their LDS bytes, active lanes stay lane-linear: %s\n", bad ? "NO" : "yes");
    hipFree(d); hipFree(o);
    return bad == 0;
}

// ---------------------------------------------------------------------------------------------- probe 2: MFMA issue rates
typedef short s4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ void __launch_bounds__(256) probe_mfma_rate(float* out, int iters) {
    hcp_f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i] = z; }
    hcp_bf16x8 a = hcp_zero8(), b = hcp_zero8();
    a[0] = (short)(0x3F80 + (threadIdx.x & 3)); b[1] = 0x3F80;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) acc[i] = hcp_mfma16(a, b, acc[i]);
            else {
                s4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
static void run_probe_mfma() {
    float* o; CK(hipMalloc(&o, 1024 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int kind = 0; kind < 2; ++kind) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (kind == 0) hipLaunchKernelGGL(probe_mfma_rate<0>, dim3(1024), dim3(256), 0, 0, o, iters);
            else hipLaunchKernelGGL(probe_mfma_rate<1>, dim3(1024), dim3(256), 0, 0, o, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        // 1024 blocks x 4 waves = 4096 waves on 1024 SIMDs = 4 waves per SIMD, each issuing iters*8 MFMAs
        const double mfma_per_simd = 4.0 * iters * 8;
        printf("[probe] %s: %.3f ms -> %.1f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz)\n", kind == 0 ? "mfma_f32_16x16x32_bf16" : "mfma_f32_16x16x16_bf16 (legacy)",
               best, best * 1e6 / mfma_per_simd, best * 1e6 / mfma_per_simd * 2.4);
    }
    hipFree(o);
}

// ---------------------------------------------------------------------------------------------- attention variants
static unsigned g_seed = 12345u;
static float frand() { g_seed = g_seed * 1664525u + 1013904223u; return (float)(g_seed >> 8) * (1.0f / 16777216.0f); }
static float nrand() { float s = 0.f; for (int i = 0; i < 12; ++i) s += frand(); return s - 6.0f; }
static unsigned short f2bf_host(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float bf2f_host(unsigned short h) { unsigned u = ((unsigned)h) << 16; float f; memcpy(&f, &u, 4); return f; }

struct Problem { int B, H, Nq, Nk, D; };

template <int D, int QT, int VAR, int NW = 4>
static void launch_v2(AttnParams p, hipStream_t st) {
    const int n = ((p.Nq + 16 * QT * NW - 1) / (16 * QT * NW)) * p.H * p.B;
    hipLaunchKernelGGL((attn2_fwd_kernel<D, QT, false, VAR, NW>), dim3(n), dim3(64 * NW), fwd_smem<D>((VAR & VAR_DEEP) ? 3 : 2), st, p);
}

template <int D, int VAR, int NW, int WPS = 2>
static void launch_v4(AttnParams p, hipStream_t st) {
    const int n = ((p.Nq + 32 * NW - 1) / (32 * NW)) * p.H * p.B;
    hipLaunchKernelGGL((attn4_fwd_kernel<D, VAR, NW, WPS>), dim3(n), dim3(64 * NW), sw_fwd_smem<D>(), st, p);
}
#define V4(D, VAR, NW, EXACT) Variant{"sw<" #D "," #VAR ",w" #NW ">", &launch_v4<D, VAR, NW>, EXACT}
#define V4P(D, VAR, NW, WPS, EXACT) Variant{"sw<" #D "," #VAR ",w" #NW ",wps" #WPS ">", &launch_v4<D, VAR, NW, WPS>, EXACT}
struct Variant;
template <int D, int VAR, int WPS, int NW = 8>
static void launch_v3(AttnParams p, hipStream_t st) {
    const int n = ((p.Nq + 32 * NW - 1) / (32 * NW)) * p.H * p.B;
    hipLaunchKernelGGL((attn3_fwd_kernel<D, VAR, WPS, NW>), dim3(n), dim3(64 * NW), pp_fwd_smem<D>(), st, p);
}
#define V3W(D, VAR, WPS, NW, EXACT) Variant{"pp<" #D "," #VAR ",wps" #WPS ",w" #NW ">", &launch_v3<D, VAR, WPS, NW>, EXACT}

struct Variant { std::string name; void (*fn)(AttnParams, hipStream_t); bool exact; };
#define V3(D, VAR, WPS, EXACT) Variant{"pp<" #D "," #VAR ",wps" #WPS ">", &launch_v3<D, VAR, WPS>, EXACT}

template <int D>
static void run_problem(const Problem& pr, const std::vector<Variant>& vars, int iters) {
    const int B = pr.B, H = pr.H, Nq = pr.Nq, Nk = pr.Nk, C = H * D;
    std::vector<unsigned short> hq((size_t)B * Nq * C), hk((size_t)B * Nk * C), hv((size_t)B * Nk * C);
    for (auto& x : hq) x = f2bf_host(nrand());
    for (auto& x : hk) x = f2bf_host(nrand());
    for (auto& x : hv) x = f2bf_host(nrand());
    // one dominating key per head in the middle of the sequence forces the online-softmax rescale path (guide §5.4 rule 26)
    if (Nk > 200) for (int c = 0; c < C; ++c) hk[(size_t)(Nk / 2 + 7) * C + c] = f2bf_host(bf2f_host(hq[(size_t)5 * C + c]) * 3.0f);
    hcp_bf16 *q, *k, *v, *o_ref, *o; float *lse_ref, *lse;
    CK(hipMalloc(&q, hq.size() * 2)); CK(hipMalloc(&k, hk.size() * 2)); CK(hipMalloc(&v, hv.size() * 2));
    CK(hipMalloc(&o_ref, hq.size() * 2)); CK(hipMalloc(&o, hq.size() * 2));
    CK(hipMalloc(&lse_ref, (size_t)B * H * Nq * 4)); CK(hipMalloc(&lse, (size_t)B * H * Nq * 4));
    CK(hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(k, hk.data(), hk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(v, hv.data(), hv.size() * 2, hipMemcpyHostToDevice));
    const float scale = 1.0f / sqrtf((float)D);
    auto product = [&](hcp_bf16* out, float* l) {
        if (hcp_attention_fwd(q, k, v, out, l, B, H, Nq, Nk, D, (long)Nq * C, C, (long)Nk * C, C, (long)Nk * C, C, (long)Nq * C, C, scale, nullptr, 0, 0, 0)) {
            printf("product kernel failed: %s\n", hcp_last_error()); exit(3);
        }
    };
    AttnParams p = {};
    p.Q = q; p.K = k; p.V = v; p.Out = o; p.lse = lse;
    p.q_bs = (long)Nq * C; p.k_bs = (long)Nk * C; p.v_bs = (long)Nk * C; p.o_bs = (long)Nq * C;
    p.q_rs = p.k_rs = p.v_rs = p.o_rs = C; p.H = H; p.Nq = Nq; p.Nk = Nk; p.scale = scale; p.B = B; p.qsplit = 1;
    product(o_ref, lse_ref);
    CK(hipDeviceSynchronize());
    std::vector<unsigned short> href(hq.size()), hout(hq.size());
    std::vector<float> hl_ref((size_t)B * H * Nq), hl((size_t)B * H * Nq);
    CK(hipMemcpy(href.data(), o_ref, href.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hl_ref.data(), lse_ref, hl_ref.size() * 4, hipMemcpyDeviceToHost));
    const double flop = 4.0 * B * H * (double)Nq * Nk * D;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("--- B%d H%d Nq%d Nk%d d%d  (%.1f GFLOP)\n", B, H, Nq, Nk, D, flop * 1e-9);
    const int nv = (int)vars.size() + 1;
    std::vector<std::vector<float>> times(nv);
    for (int round = 0; round < 4; ++round)
        for (int vi = 0; vi < nv; ++vi) {
            auto run = [&]() { if (vi == 0) product(o, lse); else vars[vi - 1].fn(p, 0); };
            if (round == 0) {
                CK(hipMemset(o, 0xff, hq.size() * 2)); CK(hipMemset(lse, 0xff, hl.size() * 4));
                run(); CK(hipDeviceSynchronize());
                hipError_t le = hipGetLastError();
                if (le != hipSuccess) { printf("%-34s launch error: %s\n", vi ? vars[vi - 1].name.c_str() : "product", hipGetErrorString(le)); continue; }
                CK(hipMemcpy(hout.data(), o, hout.size() * 2, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hl.data(), lse, hl.size() * 4, hipMemcpyDeviceToHost));
                double maxd = 0, sum = 0, maxl = 0; size_t nbad = 0;
                for (size_t i = 0; i < hout.size(); ++i) {
                    const float a = bf2f_host(hout[i]), r = bf2f_host(href[i]);
                    const double d = std::fabs((double)a - r);
                    if (!(d <= 1e30)) { ++nbad; continue; }
                    maxd = std::max(maxd, d); sum += d;
                }
                for (size_t i = 0; i < hl.size(); ++i) { const double d = std::fabs((double)hl[i] - hl_ref[i]); if (!(d <= 1e30)) ++nbad; else maxl = std::max(maxl, d); }
                printf("%-34s check vs product: max|dO| %.4f mean|dO| %.5f max|dlse| %.5f nan %zu%s\n", vi ? vars[vi - 1].name.c_str() : "product(gen1)",
                       maxd, sum / hout.size(), maxl, nbad, (vi && vars[vi - 1].exact && (maxd > 0.06 || maxl > 0.02 || nbad)) ? "   <<<<< MISMATCH" : "");
            }
            for (int w = 0; w < 2; ++w) run();
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) run();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            times[vi].push_back(ms * 1e3f / iters);
        }
    for (int vi = 0; vi < nv; ++vi) {
        if (times[vi].empty()) continue;
        std::sort(times[vi].begin(), times[vi].end());
        const float med = times[vi][times[vi].size() / 2], mn = times[vi][0];
        printf("%-34s median %8.1f us  min %8.1f us   %7.1f TFLOP/s (median)\n", vi ? vars[vi - 1].name.c_str() : "product(gen1)", med, mn, flop / med * 1e-6);
    }
    hipFree(q); hipFree(k); hipFree(v); hipFree(o_ref); hipFree(o); hipFree(lse_ref); hipFree(lse);
}

template <int D, int QT, int VAR>
static void launch_dq2(AttnParams p, hipStream_t st) {
    const int n = ((p.Nq + 64 * QT - 1) / (64 * QT)) * p.H * p.B;
    hipLaunchKernelGGL((attn2_bwd_dq_kernel<D, QT, false, VAR>), dim3(n), dim3(256), fwd_smem<D>(), st, p);
}
template <int D, int KT, int VAR>
static void launch_dkv2(AttnParams p, hipStream_t st) {
    const int n = ((p.Nk + 64 * KT - 1) / (64 * KT)) * p.H * p.B;
    hipLaunchKernelGGL((attn2_bwd_dkv_kernel<D, KT, false, VAR>), dim3(n), dim3(256), dkv_smem<D>(), st, p);
}

// Backward timings (correctness of these kernels is pytest's job: tests/test_kernels.py -m gpu): the shipped backward
// (delta + dQ + dK/dV through the C ABI) and each kernel variant alone on the same buffers.
template <int D>
static void run_bwd(const Problem& pr, const std::vector<Variant>& vars, int iters) {
    const int B = pr.B, H = pr.H, Nq = pr.Nq, Nk = pr.Nk, C = H * D;
    std::vector<unsigned short> hq((size_t)B * Nq * C), hk((size_t)B * Nk * C);
    for (auto& x : hq) x = f2bf_host(nrand());
    for (auto& x : hk) x = f2bf_host(nrand());
    hcp_bf16 *q, *k, *v, *o, *go, *dq, *dk, *dv; float *lse, *delta;
    CK(hipMalloc(&q, hq.size() * 2)); CK(hipMalloc(&go, hq.size() * 2)); CK(hipMalloc(&o, hq.size() * 2)); CK(hipMalloc(&dq, hq.size() * 2));
    CK(hipMalloc(&k, hk.size() * 2)); CK(hipMalloc(&v, hk.size() * 2)); CK(hipMalloc(&dk, hk.size() * 2)); CK(hipMalloc(&dv, hk.size() * 2));
    CK(hipMalloc(&lse, (size_t)B * H * Nq * 4)); CK(hipMalloc(&delta, (size_t)B * H * Nq * 4));
    CK(hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    for (auto& x : hq) x = f2bf_host(nrand());
    CK(hipMemcpy(go, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(k, hk.data(), hk.size() * 2, hipMemcpyHostToDevice));
    for (auto& x : hk) x = f2bf_host(nrand());
    CK(hipMemcpy(v, hk.data(), hk.size() * 2, hipMemcpyHostToDevice));
    const float scale = 1.0f / sqrtf((float)D);
    if (hcp_attention_fwd(q, k, v, o, lse, B, H, Nq, Nk, D, (long)Nq * C, C, (long)Nk * C, C, (long)Nk * C, C, (long)Nq * C, C, scale, nullptr, 0, 0, 0)) exit(3);
    AttnParams p = {};
    p.Q = q; p.K = k; p.V = v; p.O = o; p.dO = go; p.dQ = dq; p.dK = dk; p.dV = dv; p.lse = lse; p.delta = delta;
    p.q_bs = (long)Nq * C; p.k_bs = (long)Nk * C; p.v_bs = (long)Nk * C; p.o_bs = (long)Nq * C;
    p.q_rs = p.k_rs = p.v_rs = p.o_rs = C; p.H = H; p.Nq = Nq; p.Nk = Nk; p.scale = scale; p.B = B; p.qsplit = 1;
    auto product = [&]() {
        if (hcp_attention_bwd(q, k, v, o, go, lse, delta, dq, dk, dv, B, H, Nq, Nk, D, (long)Nq * C, C, (long)Nk * C, C, (long)Nk * C, C, (long)Nq * C, C,
                              scale, nullptr, 0, 0, nullptr, 0, 0)) { printf("bwd failed: %s\n", hcp_last_error()); exit(3); }
    };
    product(); CK(hipDeviceSynchronize());
    const double flop = 4.0 * B * H * (double)Nq * Nk * D;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("--- backward B%d H%d Nq%d Nk%d d%d  (fwd %.1f GFLOP; dQ kernel = 1.5x, dK/dV kernel = 2x of it in MFMA work)\n", B, H, Nq, Nk, D, flop * 1e-9);
    const int nv = (int)vars.size() + 1;
    std::vector<std::vector<float>> times(nv);
    for (int round = 0; round < 4; ++round)
        for (int vi = 0; vi < nv; ++vi) {
            auto run = [&]() { if (vi == 0) product(); else vars[vi - 1].fn(p, 0); };
            for (int w = 0; w < 2; ++w) run();
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) run();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            times[vi].push_back(ms * 1e3f / iters);
        }
    for (int vi = 0; vi < nv; ++vi) {
        std::sort(times[vi].begin(), times[vi].end());
        printf("%-34s median %8.1f us  min %8.1f us\n", vi ? vars[vi - 1].name.c_str() : "shipped bwd (delta+dQ+dKdV)", times[vi][times[vi].size() / 2], times[vi][0]);
    }
    hipFree(q); hipFree(k); hipFree(v); hipFree(o); hipFree(go); hipFree(dq); hipFree(dk); hipFree(dv); hipFree(lse); hipFree(delta);
}
#define VDQ(D, QT, VAR) Variant{"dq<" #D "," #QT "," #VAR ">", &launch_dq2<D, QT, VAR>, false}
#define VDKV(D, KT, VAR) Variant{"dkv<" #D "," #KT "," #VAR ">", &launch_dkv2<D, KT, VAR>, false}

#define V2(D, QT, VAR, EXACT) Variant{"v2<" #D "," #QT "," #VAR ">", &launch_v2<D, QT, VAR>, EXACT}
#define V2W(D, QT, VAR, NW, EXACT) Variant{"v2<" #D "," #QT "," #VAR ",w" #NW ">", &launch_v2<D, QT, VAR, NW>, EXACT}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "pmc") {          // target of tools/pmc_passes.sh: few launches, the kernels of interest only
        run_problem<40>({4, 8, 4096, 4096, 40}, {V2W(40, 2, 67, 8, true), V2W(40, 2, 66, 8, true)}, 2);   // shipped forward; variant 66 = without the XCD-aware order
        run_bwd<40>({4, 8, 4096, 4096, 40}, {}, 2);
        return 0;
    }
    const bool masked_ok = run_probe_masked();
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    {
        std::vector<Variant> vars = {
            V2W(40, 2, 67, 8, true), V2W(40, 2, 2115, 8, true), V2W(40, 2, 579, 8, false), V2W(40, 2, 2627, 8, false),   // + LSUM (2048), + PRE (512)
            V2(40, 2, 2115, true), V2(40, 2, 2627, false),
            V4P(40, 1091, 4, 3, true), V4P(40, 1603, 4, 3, false),
        };
        if (!masked_ok) printf("NOTE: masked-DMA probe failed\n");
        run_problem<40>({4, 8, 4096, 4096, 40}, vars, iters);
        run_problem<40>({2, 8, 4096 - 24, 4096 - 24, 40}, {V2(40, 2, 3, true), V2W(40, 2, 3, 8, true), V2W(40, 2, 2115, 8, true), V4(40, 67, 8, true)}, iters);     // ragged tiles
        run_problem<40>({4, 8, 4096, 77, 40}, {V2(40, 2, 3, true), V2W(40, 2, 3, 8, true), V2(40, 1, 3, true)}, iters);                  // cross-attention
    }
    run_bwd<40>({4, 8, 4096, 4096, 40}, {VDQ(40, 2, 67), VDQ(40, 2, 515), VDKV(40, 2, 67), VDKV(40, 2, 515)}, iters);   // 515 = pre-scaled Q
    run_bwd<80>({4, 8, 1024, 1024, 80}, {VDQ(80, 1, 3), VDKV(80, 1, 3)}, iters);
    run_problem<80>({4, 8, 1024, 1024, 80}, {V2(80, 1, 3, true), V2W(80, 1, 3, 8, true), V2W(80, 1, 2051, 8, true), V2(80, 2, 3, true), V4(80, 67, 4, true), V4(80, 67, 8, true)}, iters);
    run_problem<160>({4, 8, 256, 256, 160}, {V2(160, 1, 3, true)}, iters);
    run_problem<64>({2, 10, 4096, 4096, 64}, {V2(64, 2, 3, true), V2(64, 2, 2051, true), V2W(64, 2, 3, 8, true), V2W(64, 1, 3, 8, true), V4(64, 67, 4, true), V4(64, 67, 8, true)}, iters);
    return 0;
}
