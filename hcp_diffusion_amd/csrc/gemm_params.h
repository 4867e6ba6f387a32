// gemm_params.h — the launch descriptor shared by the GEMM / implicit-conv translation units (gemm.hip, gemm_pp.hip).
#pragma once
#include "hcp_common.h"

namespace hcp_gemm {

struct ConvDesc {
    const hcp_bf16* X1; int C1;   // first source tensor  [B, Hs, Ws, C1]
    const hcp_bf16* X2; int C2;   // optional second source (channel concat), else null/0
    int Hs, Ws;                   // source spatial dims (memory)
    int Ho, Wo;                   // output spatial dims (rows of the implicit A matrix)
    int stride;                   // 1 or 2
    int up;                       // 1: source is nearest-upsampled 2x before the conv (fwd only)
    int pad;                      // 1: taps -1..+1 (padding 1); 0: taps 0..+2 (F.pad(0,1,0,1) + padding 0: the VAE encoder's Downsample2D), fwd only
};

struct GemmParams {
    const hcp_bf16* A; int lda;
    const hcp_bf16* A2; int lda2; int K2;
    const hcp_bf16* B; int ldb;
    const hcp_bf16* B2; int ldb2;
    int M, N, K;
    void* D; int ldd; int out_f32;
    const float* bias;
    const float* rowbias; int rowbias_ld; int rows_per_group;
    const hcp_bf16* residual; int ldr;
    float alpha;
    int tiles_m;
    int nsplit; int kt_per_split;   // split-K over the primary K tiles (grid.y)
    float* slabs;                   // [nsplit][M][N] fp32 partials when nsplit > 1
    // fused LoRA (LORA kernels): T = A L^T is accumulated next to the main tile from the same A tiles, rounded to
    // bf16, then D += T E^T as one extra k-step.  L [32,K] (ldl = K), E [N,32], Tout [M,32] (optional, for wgrad).
    const hcp_bf16* L; const hcp_bf16* E; hcp_bf16* Tout;
    int loaders;                    // 1: launch the loader-wave variant of the v2 kernel where one is instantiated (dispatch table / tools)
    int dbg;                        // tools/ablate_gemm.py: 1 = skip the DMA after the first tile, 2 = skip the MFMAs, 4 = skip LDS reads + MFMAs
    ConvDesc cv;
};

constexpr int BK = 64;

// gemm_pp.hip — the ping-pong main loop (two compute groups half a phase apart + 4 loader waves).  `p` arrives with tiles_m,
// nsplit, kt_per_split and slabs set by the dispatcher; ring = depth of the LDS ring (2..4, lowered to what fits 160 KB).
// Returns -2 when no kernel is instantiated for (bm, bn, mode, lora) — the caller then keeps its own kernels.
int gemm_pp_launch(GemmParams& p, int bm, int bn, int mode, bool lora, int ring, hipStream_t stream);

}  // namespace hcp_gemm
