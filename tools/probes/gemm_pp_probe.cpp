// tools/probes/gemm_pp_probe.cpp — GPU-only A/B of the GEMM / conv main loops through the C ABI of libhcp_mi355x_tools.so
// (no Python, no torch: starts in seconds on a fresh box).  For every shape: the dispatched kernel (tables as shipped), then forced
// variants (hcp_debug_set_gemm_config / _loaders), each timed COLD-ish (rotating over NSETS distinct operand sets > the L2s) with
// hipEvents around REPS launches, interleaved over ROUNDS rounds (median and min reported); outputs of every variant are compared
// with the first variant's (max relative difference) and re-run once to screen for run-to-run differences (races).
//   build:  hipcc -O2 -std=c++17 -DHCP_TOOLS tools/probes/gemm_pp_probe.cpp -o tools/probes/gemm_pp_probe -Lhcp_diffusion_amd -lhcp_mi355x_tools -Wl,-rpath,'$ORIGIN/../../hcp_diffusion_amd'
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hcp_mi355x_tools.h"      // the ABI as shipped (this file used to restate the prototypes and went stale twice)

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static float urand() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (float)((rng >> 40) & 0xffffff) / 8388608.0f - 1.0f; }

static void* dev_random_bf16(size_t n, float scale) {
    std::vector<uint16_t> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = f2bf(urand() * scale);
    void* d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}
static float* dev_random_f32(size_t n) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = urand();
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

struct Shape { const char* name; int kind; int M, N, K; int B, H, C; int res; };   // kind 0 gemm, 1 lora, 2 conv fwd, 3 conv dgrad
struct Variant { const char* name; int cfg; int loaders; int abl; };

int main(int argc, char** argv) {
    const int NSETS = 6, REPS = 24, ROUNDS = 5;
    const char* only = argc > 1 && strcmp(argv[1], "-") ? argv[1] : nullptr;
    const bool ablate = argc > 2 && !strcmp(argv[2], "abl");
    std::vector<Shape> shapes = {
        {"conv C320 @64x64 B4 (fwd)", 2, 16384, 320, 2880, 4, 64, 320, 0},
        {"conv C320 @64x64 B4 (dgrad)", 3, 16384, 320, 2880, 4, 64, 320, 0},
        {"conv C320 @64x64 B4 fwd +residual", 2, 16384, 320, 2880, 4, 64, 320, 1},
        {"conv C320 @64x64 B4 fwd +rowbias", 2, 16384, 320, 2880, 4, 64, 320, 2},
        {"conv C320 @64x64 B4 fwd cold30 +rowbias", 2, 16384, 320, 2880, 4, 64, 320, 2},
        {"conv C320 @64x64 B4 fwd cold30", 2, 16384, 320, 2880, 4, 64, 320, 0},
        {"gemm M16384 N320 K2880", 0, 16384, 320, 2880, 0, 0, 0, 0},
        {"gemm M16384 N320 K320 +res", 0, 16384, 320, 320, 0, 0, 0, 1},
        {"lora M16384 N320 K320 +res", 1, 16384, 320, 320, 0, 0, 0, 1},
        {"lora M16384 N960 K320", 1, 16384, 960, 320, 0, 0, 0, 0},
        {"lora M16384 N2560 K320", 1, 16384, 2560, 320, 0, 0, 0, 0},
        {"lora M16384 N320 K1280 +res", 1, 16384, 320, 1280, 0, 0, 0, 1},
        {"conv C640 @32x32 B4 (fwd)", 2, 4096, 640, 5760, 4, 32, 640, 0},
        {"lora M4096 N5120 K640", 1, 4096, 5120, 640, 0, 0, 0, 0},
        {"lora M4096 N640 K2560 +res", 1, 4096, 640, 2560, 0, 0, 0, 1},
        {"lora M4096 N640 K640 +res", 1, 4096, 640, 640, 0, 0, 0, 1},
        {"conv C1280 @16x16 B4 (fwd)", 2, 1024, 1280, 11520, 4, 16, 1280, 0},
        {"lora M1024 N1280 K1280 +res", 1, 1024, 1280, 1280, 0, 0, 0, 1},
        {"lora M1024 N10240 K1280", 1, 1024, 10240, 1280, 0, 0, 0, 0},
        {"lora M2048 N1280 K1280 +res (SDXL)", 1, 2048, 1280, 1280, 0, 0, 0, 1},
        {"lora M8192 N640 K640 +res (SDXL)", 1, 8192, 640, 640, 0, 0, 0, 1},
        {"lora M2048 N1280 K1280 +res (SDXL) cold30", 1, 2048, 1280, 1280, 0, 0, 0, 1},
        {"lora M4096 N640 K2560 +res cold30", 1, 4096, 640, 2560, 0, 0, 0, 1},
        {"lora M1024 N1280 K1280 +res cold30", 1, 1024, 1280, 1280, 0, 0, 0, 1},
        {"lora M16384 N320 K1280 +res cold30", 1, 16384, 320, 1280, 0, 0, 0, 1},
        {"gemm 4096^3", 0, 4096, 4096, 4096, 0, 0, 0, 0},
        {"gemm 8192^3", 0, 8192, 8192, 8192, 0, 0, 0, 0},
    };
    std::vector<Variant> variants = {
        {"dispatched", -1, -1, 0},
        {"v2 128x160 ring4", 13 + 16, 4, 0},
        {"pp 128x160 ring4", 13 + 16, 12, 0},
        {"pp 128x160 ring3", 13 + 16, 11, 0},
        {"pp 64x160 ring4", 14 + 16, 12, 0},
        {"pp 128x128 ring4", 15 + 16, 12, 0},
        {"pp 128x160 s2 ring4", 13 + 32, 12, 0},
        {"pp 64x160 s2 ring4", 14 + 32, 12, 0},
    };
    if (ablate) variants = {
        {"v2 128x160 ring4", 13 + 16, 4, 0}, {"v2 no stores", 13 + 16, 4, 64}, {"v2 no DMA", 13 + 16, 4, 8}, {"v2 no reads no MFMA", 13 + 16, 4, 48}, {"v2 no MFMA", 13 + 16, 4, 16},
        {"pp 128x160 ring4", 13 + 16, 12, 0}, {"pp no A", 13 + 16, 12, 0x100}, {"pp no B", 13 + 16, 12, 0x200}, {"pp no DMA", 13 + 16, 12, 0x300},
        {"pp no MFMA", 13 + 16, 12, 0x400}, {"pp barriers only", 13 + 16, 12, 0x800}, {"pp no DMA no MFMA", 13 + 16, 12, 0x700},
        {"pp nothing", 13 + 16, 12, 0xb00}, {"pp no A no MFMA", 13 + 16, 12, 0x500}, {"pp no B no MFMA", 13 + 16, 12, 0x600},
        {"pp no setprio", 13 + 16, 12, 0x1000}, {"pp loaders prio 3", 13 + 16, 12, 0x2000}, {"pp no setprio, ld 3", 13 + 16, 12, 0x3000},
        {"pp ring3 no setprio", 13 + 16, 11, 0x1000}, {"pp ring3 ld prio 3", 13 + 16, 11, 0x2000},
    };
    hipStream_t st; CK(hipStreamCreate(&st));
    size_t ws_bytes = (size_t)512 << 20;
    void* ws; CK(hipMalloc(&ws, ws_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    for (const Shape& s : shapes) {
        if (only && !strstr(s.name, only)) continue;
        const bool big = (double)s.M * s.N * s.K > 1e11;
        const int nsets = big ? 2 : strstr(s.name, "cold30") ? 30 : NSETS, reps = big ? 6 : strstr(s.name, "cold30") ? 60 : REPS;
        std::vector<void*> A(nsets), Bw(nsets), D(nsets), R(nsets);
        void *Lw = nullptr, *Ew = nullptr, *T = nullptr;
        const size_t a_elems = s.kind >= 2 ? (size_t)s.B * s.H * s.H * s.C : (size_t)s.M * s.K;
        for (int i = 0; i < nsets; ++i) {
            A[i] = dev_random_bf16(a_elems, 1.0f);
            Bw[i] = dev_random_bf16((size_t)s.N * s.K, 0.05f);
            CK(hipMalloc(&D[i], (size_t)s.M * s.N * 2));
            R[i] = s.res == 1 ? dev_random_bf16((size_t)s.M * s.N, 1.0f) : nullptr;
        }
        float* bias = dev_random_f32(s.N);
        float* rowbias = s.res == 2 ? dev_random_f32((size_t)16 * s.N) : nullptr;
        if (s.kind == 1) { Lw = dev_random_bf16((size_t)32 * s.K, 0.05f); Ew = dev_random_bf16((size_t)s.N * 32, 0.05f); CK(hipMalloc(&T, (size_t)s.M * 32 * 2)); }
        auto run = [&](int i) -> int {
            if (s.kind == 0) return hcp_gemm_bf16(A[i], s.K, Bw[i], s.K, D[i], s.N, s.M, s.N, s.K, nullptr, 0, nullptr, 0, 0, bias, nullptr, 0, 1, R[i], s.N, nullptr, nullptr, nullptr, 1.0f, 0, ws, ws_bytes, st);
            if (s.kind == 1) return hcp_gemm_lora_bf16(A[i], s.K, Bw[i], s.K, Lw, Ew, T, 32, D[i], s.N, s.M, s.N, s.K, bias, R[i], s.N, nullptr, nullptr, nullptr, ws, ws_bytes, st);
            return hcp_conv3x3_bf16(A[i], s.C, nullptr, 0, s.B, s.H, s.H, s.H, s.H, s.kind == 2 ? 0 : 1, 1, 0, 1, Bw[i], s.N, D[i], s.N, bias, rowbias, s.N, R[i], s.N, 0, nullptr, nullptr, ws, ws_bytes, st);
        };
        const double flop = 2.0 * s.M * s.N * (s.K + (s.kind == 1 ? 32 : 0));
        printf("== %s  (%.1f GFLOP)\n", s.name, flop / 1e9);
        std::vector<uint16_t> ref((size_t)s.M * s.N), got((size_t)s.M * s.N), again((size_t)s.M * s.N);
        std::vector<std::vector<double>> times(variants.size());
        std::vector<int> ok(variants.size(), 1);
        for (size_t v = 0; v < variants.size(); ++v) {
            hcp_debug_set_gemm_config(variants[v].cfg); hcp_debug_set_gemm_loaders(variants[v].loaders); hcp_debug_set_gemm_ablation(variants[v].abl);
            CK(hipMemsetAsync(D[0], 0, (size_t)s.M * s.N * 2, st));
            if (run(0) != 0) { ok[v] = 0; printf("   %-22s refused: %s\n", variants[v].name, hcp_last_error()); continue; }
            hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) { printf("   %-22s FAILED: %s\n", variants[v].name, hipGetErrorString(e)); return 2; }
            CK(hipMemcpy(got.data(), D[0], got.size() * 2, hipMemcpyDeviceToHost));
            if (v == 0) ref = got;
            double num = 0, den = 0;
            for (size_t i = 0; i < got.size(); ++i) { double d = bf2f(got[i]) - bf2f(ref[i]); num += d * d; den += (double)bf2f(ref[i]) * bf2f(ref[i]); }
            int races = 0;
            for (int r = 0; r < 3; ++r) {
                CK(hipMemsetAsync(D[0], 0, (size_t)s.M * s.N * 2, st));
                run(0); CK(hipStreamSynchronize(st));
                CK(hipMemcpy(again.data(), D[0], again.size() * 2, hipMemcpyDeviceToHost));
                if (memcmp(again.data(), got.data(), got.size() * 2) != 0) ++races;
            }
            printf("   %-22s rel diff vs dispatched %.2e  reruns differing %d/3\n", variants[v].name, std::sqrt(num / (den + 1e-30)), races);
        }
        for (int round = 0; round < ROUNDS; ++round)
            for (size_t v = 0; v < variants.size(); ++v) {
                if (!ok[v]) continue;
                hcp_debug_set_gemm_config(variants[v].cfg); hcp_debug_set_gemm_loaders(variants[v].loaders); hcp_debug_set_gemm_ablation(variants[v].abl);
                for (int i = 0; i < 2; ++i) run(i % nsets);
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < reps; ++i) run(i % nsets);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                times[v].push_back(ms * 1e3 / reps);
            }
        if (argc > 2 && !strcmp(argv[2], "icache")) {          // the same launch timed alone (events around ONE launch) after itself vs after other kernels
            hcp_debug_set_gemm_config(-1); hcp_debug_set_gemm_loaders(-1); hcp_debug_set_gemm_ablation(0);
            void* xa = dev_random_bf16((size_t)4096 * 640, 1.0f); void* xb = dev_random_bf16((size_t)5120 * 640, 0.05f); void* xd; CK(hipMalloc(&xd, (size_t)4096 * 5120 * 2));
            void* xl = dev_random_bf16((size_t)32 * 640, 0.05f); void* xe = dev_random_bf16((size_t)5120 * 32, 0.05f); void* xt; CK(hipMalloc(&xt, (size_t)4096 * 32 * 2));
            auto others = [&]() {                             // three other kernel templates (fused-LoRA 128x160 v2, plain 64x160, split-K + reduce)
                hcp_gemm_lora_bf16(xa, 640, xb, 640, xl, xe, xt, 32, xd, 640, 4096, 640, 640, nullptr, nullptr, 0, nullptr, nullptr, nullptr, ws, ws_bytes, st);
                hcp_gemm_bf16(xa, 640, xb, 640, xd, 1280, 1024, 1280, 640, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0, 1, nullptr, 0, nullptr, nullptr, nullptr, 1.0f, 0, ws, ws_bytes, st);
                hcp_gemm_bf16(xa, 640, xb, 640, xd, 320, 256, 320, 640, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, 0, 1, nullptr, 0, nullptr, nullptr, nullptr, 1.0f, 0, ws, ws_bytes, st);
            };
            for (int mode = 0; mode < 2; ++mode) {
                std::vector<float> ts;
                for (int i = 0; i < 60; ++i) {
                    if (mode == 1) others(); else run((i + 1) % nsets);
                    CK(hipEventRecord(e0, st));
                    run(i % nsets);
                    CK(hipEventRecord(e1, st));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    ts.push_back(ms * 1e3f);
                }
                std::sort(ts.begin(), ts.end());
                printf("   single launch between two events, %s: median %.1f us  min %.1f us\n", mode ? "after three OTHER kernels" : "after ITSELF", ts[ts.size() / 2], ts[0]);
            }
        }
        if (argc > 2 && !strcmp(argv[2], "sustain")) {         // does the per-launch time drift under a SUSTAINED load (clock / power management)?
            hcp_debug_set_gemm_config(-1); hcp_debug_set_gemm_loaders(-1); hcp_debug_set_gemm_ablation(0);
            printf("   sustained (dispatched), us per launch in consecutive chunks of 400 launches:");
            for (int c = 0; c < 12; ++c) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < 400; ++i) run(i % nsets);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf(" %.1f", ms * 1e3 / 400);
            }
            printf("\n");
        }
        for (size_t v = 0; v < variants.size(); ++v) {
            if (!ok[v]) continue;
            std::sort(times[v].begin(), times[v].end());
            const double med = times[v][times[v].size() / 2], mn = times[v][0];
            printf("   %-22s median %8.1f us  min %8.1f us  %7.1f TFLOP/s\n", variants[v].name, med, mn, flop / med / 1e6);
        }
        fflush(stdout);
        for (int i = 0; i < nsets; ++i) { CK(hipFree(A[i])); CK(hipFree(Bw[i])); CK(hipFree(D[i])); if (R[i]) CK(hipFree(R[i])); }
        CK(hipFree(bias)); if (Lw) { CK(hipFree(Lw)); CK(hipFree(Ew)); CK(hipFree(T)); }
    }
    hcp_debug_set_gemm_config(-1); hcp_debug_set_gemm_loaders(-1); hcp_debug_set_gemm_ablation(0);
    return 0;
}
